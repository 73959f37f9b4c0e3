"""Imports the package directory `fpga-mpeg2-encoder_amd/` (not a valid Python identifier)
under the module name `fpga_mpeg2_encoder_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_NAME = "fpga_mpeg2_encoder_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    pkg = os.path.join(_ROOT, "fpga-mpeg2-encoder_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg, "__init__.py"),
                                                  submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
