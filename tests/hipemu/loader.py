"""TEST INFRASTRUCTURE: builds and opens libmds_emu.so, the host simulator build of csrc/*.hip
(see hipemu.h).  Only the test-suite uses this; the product loader (mds.cabi.load) cannot."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "ball-action-spotting_amd", "csrc")
_LIB = None


def load_emulator():
    global _LIB
    if _LIB is None:
        subprocess.run(["make", "-s", "-j8", "emu"], cwd=CSRC, check=True)
        from mds.cabi import Lib
        _LIB = Lib(os.path.join(HERE, "libmds_emu.so"))
    return _LIB
