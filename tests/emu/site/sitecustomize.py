# gfx950emu harness: a child process of an emulated test run (GFX950EMU=1 in its environment, this directory on its PYTHONPATH)
# gets the stand-in HIP runtime and the stand-in torch before its own code runs.  Does nothing otherwise.
import os
if os.environ.get('GFX950EMU') == '1':
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import emurun
    emurun.load()
