"""Where the end-to-end fit of examples/fit_network.py spends its host time: second run in the process under cProfile.
usage: python tools/experiments/r04_fit_profile.py [scale] [builtin|device]"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'examples'))
import fit_network

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
remesher = sys.argv[2] if len(sys.argv) > 2 else 'device'
fit_network.main(scale, remesher)
t0 = time.time()
fit_network.main(scale, remesher)
print('second run, whole main(): %.2f s' % (time.time() - t0))
pr = cProfile.Profile()
pr.enable()
fit_network.main(scale, remesher)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
