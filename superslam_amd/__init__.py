"""superslam_amd - MI355X (gfx950) deep-feature front-end for SuperSLAM.

Python host layer over libsuperslam_hip.so (C ABI: include/sship.h).  Importing this package never
imports the CPU oracle; the HIP library is the only compute path.
"""
from . import _lib  # noqa: F401
from .eigenplaces import EigenPlaces  # noqa: F401
from .frontend import FrontEndBatch, process_stereo  # noqa: F401
from .lightglue import LightGlue, LightGlueEngine, MatchResult  # noqa: F401
from .pool import DescriptorPool, DeviceDescriptors  # noqa: F401
from .superpoint import Features, SuperPoint  # noqa: F401

__all__ = ["SuperPoint", "LightGlue", "LightGlueEngine", "MatchResult", "Features", "DescriptorPool",
           "DeviceDescriptors", "FrontEndBatch", "process_stereo", "EigenPlaces"]
