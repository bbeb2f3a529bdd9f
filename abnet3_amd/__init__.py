"""abnet3_amd: MI355X-native implementation of bootphon/abnet3's Siamese
training hot path (SiameseNetwork forward/backward, coscos2/cosmargin, DTW
frame alignment, filterbanks) behind the reference's own class surfaces.

Submodules mirror the reference's module names so that the class-name lookup
of abnet3/gridsearch.py:145-202 resolves against this package unchanged:
    abnet3_amd.model, .loss, .trainer, .embedder, .dataloader, .features, .utils
"""
__version__ = '0.1.0'
