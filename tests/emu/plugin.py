"""pytest plugin of the gfx950emu harness (test infrastructure): `python -m pytest -p tests.emu.plugin -m gpu tests/test_gpu_encode.py`
runs GPU tests on the ISA-level emulator instead of a GPU: the stand-in HIP runtime is loaded in front of libflacgpu.so, a stand-in
torch (numpy arrays over the emulator's memory) takes torch's name, and child processes the tests start inherit both."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import emurun  # noqa: E402

emurun.load()
os.environ['GFX950EMU'] = '1'
os.environ['PYTHONPATH'] = os.path.join(HERE, 'site') + os.pathsep + os.environ.get('PYTHONPATH', '')
