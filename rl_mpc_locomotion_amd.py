"""Import shim: the package directory is named ``rl-mpc-locomotion_amd`` (hyphenated, as the project
layout prescribes), which is not a valid Python identifier.  ``import rl_mpc_locomotion_amd`` loads
that directory as a regular package under the importable name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rl-mpc-locomotion_amd")
_spec = importlib.util.spec_from_file_location(
    "rl_mpc_locomotion_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rl_mpc_locomotion_amd"] = _mod
_spec.loader.exec_module(_mod)
