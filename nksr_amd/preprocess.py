"""``nksr.get_estimate_normal_preprocess_fn(knn, deg)`` (reference call sites
examples/recons_waymo.py:36, gis_app.py:41; CPU recipe examples/recons_waymo_cpu.py:21-41):
kNN-PCA normals, flipped towards the sensor, grazing (> deg) points dropped.
SURVEY.md section 8(f)-1 ranks this "next"; the HIP implementation lands after the hot path."""


def get_estimate_normal_preprocess_fn(knn=64, deg=85.0):
    def fn(xyz, normal, sensor):
        from .normals import estimate_normals_knn
        return estimate_normals_knn(xyz, normal, sensor, int(knn), float(deg))
    return fn
