"""`simple_knn._C`: the one function the reference uses."""
from humangaussian_amd.knn import distCUDA2  # noqa: F401
