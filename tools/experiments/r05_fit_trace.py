"""Three end-to-end fits (examples/fit_network.py 0.2, remesher on the device) in one process, for a kernel trace:
rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/experiments/r05_fit_trace.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'examples'))
import fit_network
for k in range(3):
    fit_network.main(0.2, 'device')
