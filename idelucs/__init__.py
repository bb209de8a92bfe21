"""`idelucs` -- the reference package's name, served by the MI355X implementation: `import idelucs`, `from idelucs.cluster import
iDeLUCS_cluster`, `from idelucs.utils import kmersFasta`, `python -m idelucs ...` resolve to idelucs_amd, so callers written against
Kari-Genomics-Lab/iDeLUCS (idelucs/__init__.py:3-8, pyproject.toml:16) run unchanged.  Nothing is implemented here."""
import importlib
import sys

import idelucs_amd
from idelucs_amd import *                                    # noqa: F401,F403  (the reference's __all__ = the sub-modules)
from idelucs_amd import (IID_loss, IID_model, AugmentFasta, SequenceDataset, SummaryFasta, cgr, cgrFasta, check_sequence,  # noqa: F401
                         cluster_acc, create_dataloader, iDeLUCS_cluster, info_nce_loss, kmer_counts, kmer_rev_comp, kmersFasta,
                         reverse_complement)
from idelucs_amd import LossFunctions, PytorchUtils, cluster, kmers, models, utils   # noqa: F401

__version__ = idelucs_amd.__version__
__all__ = idelucs_amd.__all__
for _name in ("utils", "kmers", "models", "cluster", "LossFunctions", "PytorchUtils", "posthoc", "dist", "fused", "training"):
    sys.modules[f"{__name__}.{_name}"] = importlib.import_module(f"idelucs_amd.{_name}")
