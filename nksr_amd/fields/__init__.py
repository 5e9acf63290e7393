from .base_field import BaseField, EvaluationResult, MeshingResult
from .kernel_field import KernelField, load_field, save_field
from .mask_fields import LayerField, NeuralField, PCNNField

__all__ = ['BaseField', 'EvaluationResult', 'MeshingResult', 'KernelField', 'save_field', 'load_field', 'LayerField', 'NeuralField', 'PCNNField']
