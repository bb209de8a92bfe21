"""idelucs_amd -- MI355X-native implementation of the iDeLUCS hot path (k-mer/CGR vectoriser,
mimic augmentation, contrastive-IIC training epoch) behind the reference's Python surface
(reference idelucs/__init__.py:1-17).  Importing it requires the built HIP library
(idelucs_amd/csrc/libidelucs_hip.so); there is no CPU fallback.
"""
__version__ = (1, 2, 6)

from .utils import (check_sequence, SummaryFasta, reverse_complement, kmer_rev_comp, kmersFasta, cgrFasta,
                    cluster_acc, SequenceDataset, AugmentFasta, create_dataloader)
from .kmers import kmer_counts, cgr
from .models import IID_model
from .LossFunctions import IID_loss, info_nce_loss
from .cluster import iDeLUCS_cluster

__all__ = ["utils", "kmers", "models", "cluster", "LossFunctions"]
