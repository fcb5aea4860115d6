"""PMC aid (not product): launches of one halo kernel generation on one shape, for rocprofv3 --pmc.   python tools/v3pmc.py old|v3 C H [kind]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tools"))
import v3check  # noqa: E402

which, C_, Hh = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "stats"
v3check.run(kind, 64, C_, Hh, which == "v3", 5)
