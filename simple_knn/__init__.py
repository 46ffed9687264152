"""Drop-in module name for the reference: `from simple_knn._C import distCUDA2`
(/root/reference/gaussiansplatting/scene/gaussian_model.py:20, gs_renderer.py:14) resolves here
when this repository is on sys.path; implemented in HIP (humangaussian_amd/csrc/knn.hip)."""
