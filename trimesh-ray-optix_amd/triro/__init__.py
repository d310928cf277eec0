"""triro (MI355X build): `trimesh.ray`-style ray/mesh queries on AMD gfx950.

Same import path and public surface as lcp29/trimesh-ray-optix 1.3.1
(`from triro.ray.ray_optix import RayMeshIntersector`, triro/__init__.py:2); the OptiX
backend is replaced by libtriro_hip.so (hand-written HIP LBVH builder + stackless traversal).
"""
__version__ = "1.3.1+mi355x.0"
