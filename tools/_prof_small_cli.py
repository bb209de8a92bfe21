import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs("/tmp/w", exist_ok=True); os.chdir("/tmp/w")
t00 = time.time()
from idelucs_amd.__main__ import main as cli
argv = ["--sequence_file", root + "/tests/data/Influenza-A.fas", "--GT_file", root + "/tests/data/Influenza-A_GT.tsv", "--n_clusters", "5", "--k", "6"]
t0 = time.time(); cli(argv); print("IDELUCS_TUNABLEOP=%s: import %.2f s, first run wall %.2f s" % (os.environ.get("IDELUCS_TUNABLEOP", "auto"), t0 - t00, time.time() - t0))
