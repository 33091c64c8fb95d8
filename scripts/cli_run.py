"""The shipped loop (model.run() as bench.py --cli drives it) for N epochs: prints its figures.  For kernel traces:
   cd /tmp && rocprofv3 --kernel-trace --output-format rocpd -d /tmp/cli -o p -- python scripts/cli_run.py 100; python scripts/rocpd_gaps.py /tmp/cli/*/*.db"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.run_cli(epochs=int(sys.argv[1]) if len(sys.argv) > 1 else 100)
print({k: v for k, v in r.items() if k != "cli_what"})
