"""gfx950emu harness: run one of the GPU-box tools (tests/tools/gpu_fuzz.py ...) on the emulator instead of a GPU.
usage: python tests/emu/run_tool.py tests/tools/gpu_fuzz.py 6000000 200"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import emurun  # noqa: E402

emurun.load()
os.environ['GFX950EMU'] = '1'
os.environ['PYTHONPATH'] = os.path.join(HERE, 'site') + os.pathsep + os.environ.get('PYTHONPATH', '')
tool = sys.argv[1]
sys.argv = sys.argv[1:]
runpy.run_path(tool, run_name='__main__')
