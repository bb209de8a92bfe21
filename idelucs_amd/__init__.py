"""idelucs_amd -- MI355X-native implementation of the iDeLUCS hot path (k-mer/CGR vectoriser, mimic
augmentation, contrastive-IIC training epoch) behind the reference's Python surface.  Importing it needs
the built HIP library (idelucs_amd/csrc/libidelucs_hip.so); there is no CPU fallback.

The names below are the ones the reference package exports (idelucs/__init__.py:3-8), so
`import idelucs_amd as idelucs` is a drop-in for the hot path.
"""
__version__ = (1, 2, 6)

from . import utils, kmers, models, cluster, LossFunctions  # noqa: F401  (sub-modules, as in the reference's __all__)

# k-mer / CGR counters (the reference's Cython module)
kmer_counts, cgr = kmers.kmer_counts, kmers.cgr
# data layer
check_sequence, SummaryFasta = utils.check_sequence, utils.SummaryFasta
reverse_complement, kmer_rev_comp = utils.reverse_complement, utils.kmer_rev_comp
kmersFasta, cgrFasta, AugmentFasta = utils.kmersFasta, utils.cgrFasta, utils.AugmentFasta
SequenceDataset, create_dataloader, cluster_acc = utils.SequenceDataset, utils.create_dataloader, utils.cluster_acc
# training
IID_model = models.IID_model
IID_loss, info_nce_loss = LossFunctions.IID_loss, LossFunctions.info_nce_loss
iDeLUCS_cluster = cluster.iDeLUCS_cluster

__all__ = ["utils", "kmers", "models", "cluster", "LossFunctions"]
