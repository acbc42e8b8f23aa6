"""wiki-grx-gym_amd: MI355X-native GR1T1/GR1T2 environment step behind the reference's
task_registry / VecEnv surface.  See DESIGN.md (repo root) for the path, its boundary and kernels.

Import as ``wiki_grx_gym_amd`` (shim package next to this directory)."""
__version__ = "0.1.0"
