"""cProfile of the shipped training loop (model.run() as bench.py --cli drives it): where the HOST time of a minibatch goes."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pr = cProfile.Profile()
pr.enable()
r = bench.run_cli(epochs=int(sys.argv[1]) if len(sys.argv) > 1 else 60)
pr.disable()
print({k: v for k, v in r.items() if k != "cli_what"})
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print("\n".join(l[:170] for l in s.getvalue().splitlines()[:75]))
