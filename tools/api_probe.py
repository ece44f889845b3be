"""api_e2e leg of bench.py alone (drop-in classes on host buffers)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.api_e2e(float(sys.argv[1]) if len(sys.argv) > 1 else 600.0, 48000), indent=1))
