from .base_field import BaseField, EvaluationResult, MeshingResult
from .kernel_field import KernelField
from .mask_fields import LayerField, NeuralField, PCNNField

__all__ = ['BaseField', 'EvaluationResult', 'MeshingResult', 'KernelField', 'LayerField', 'NeuralField', 'PCNNField']
