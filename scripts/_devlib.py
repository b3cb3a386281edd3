"""Developer / test plumbing, NOT product: lets an A/B launcher hand a variant build of the C-ABI library to a child interpreter.

The product package never reads an environment variable to pick its library (superslam_amd/_lib.py; include/sship.h "Environment").  The A/B
tests and the scripts under scripts/ run every kernel-selection mode in its own interpreter and name the build to load in SSHIP_DEV_LIBRARY;
each such script calls `use_dev_library()` FIRST, which turns that variable into an explicit superslam_amd._lib.set_library_path() call.
A process that does not call this function (a SuperSLAM host, bench.py without --library, the parity tests) is unaffected by the variable."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def use_dev_library(path=None):
    from superslam_amd import _lib

    path = path or os.environ.get("SSHIP_DEV_LIBRARY")
    if path:
        _lib.set_library_path(path)
    return _lib.LIB_PATH
