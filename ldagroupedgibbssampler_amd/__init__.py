"""MI355X-native Grouped Gibbs Sampler for LDA: the z-sampling hot path of
cc.mallet.topics.LDAGroupedGibbsSampler as hand-written HIP behind a C-ABI
(include/ggs_hip.h).  See DESIGN.md."""
__version__ = "0.1.0"
