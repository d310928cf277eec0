"""Import-time bring-up, mirroring triro/ray/__init__.py:18-22 of the reference.

The reference JIT-compiles its extension and initialises OptiX here (init_optix,
create_optix_context, create_optix_module, create_optix_pipelines, build_sbts).  The HIP
backend keeps those five entry points (they collapse into loading the prebuilt C-ABI library
and, when a GPU is visible, creating the per-device runtime table).  Importing never fails
on a machine without a GPU -- the first query does, loudly.
"""
import triro.backend.ops as hops

hops.init_optix()
hops.create_optix_context()
hops.create_optix_module()
hops.create_optix_pipelines()
hops.build_sbts()
