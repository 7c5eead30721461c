"""Global configuration knob set, mirroring the reference's `cfg` AttrDict (reference config.py:46-93,
utils/attr_dict.py).  The only knob the reference's hot path reads is cfg.MODEL.BNFUNC
(config.py:92-93, used by mynn.Norm2d); the build adds the activation dtype and the conv backend.
"""
import torch


class AttrDict(dict):
    """dict with attribute access and an immutability switch (reference utils/attr_dict.py)."""

    IMMUTABLE = "__immutable__"

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__[AttrDict.IMMUTABLE] = False

    def __getattr__(self, name):
        if name in self.__dict__:
            return self.__dict__[name]
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[AttrDict.IMMUTABLE]:
            raise AttributeError('Attempted to set "%s" to "%s", but AttrDict is immutable' % (name, value))
        if name in self.__dict__:
            self.__dict__[name] = value
        else:
            self[name] = value

    def immutable(self, is_immutable):
        self.__dict__[AttrDict.IMMUTABLE] = is_immutable
        for v in list(self.__dict__.values()) + list(self.values()):
            if isinstance(v, AttrDict):
                v.immutable(is_immutable)

    def is_immutable(self):
        return self.__dict__[AttrDict.IMMUTABLE]


cfg = AttrDict()
cfg.ITER = 0
cfg.EPOCH = 0
cfg.MODEL = AttrDict()
cfg.MODEL.BN = "hip-batchnorm"
# set lazily by network.mynn (the HIP BatchNorm2d subclass); the reference default is
# torch.nn.SyncBatchNorm, which on one process is plain batch statistics.
cfg.MODEL.BNFUNC = None
# activation storage dtype of the HIP path: torch.float32 (parity runs) or torch.bfloat16 (bench)
cfg.MODEL.ACT_DTYPE = torch.float32
# training: fuse final bilinear upsample + cross entropy (the full-resolution logits are never written)
cfg.MODEL.FUSE_UPSAMPLE_CE = True
# MRFP+ head: evaluate final2(Upsample(dec1) + OCout_dec) as Upsample(final2(dec1)) + final2(OCout_dec) (a 1x1 conv
# commutes with bilinear interpolation): the 2x upsample runs on the class scores, not on 256 channels
cfg.MODEL.COMMUTE_O2 = __import__("os").environ.get("MRFP_COMMUTE_O2", "1") != "0"
# decoder input cat([bot_fine(low level) 48, Upsample(bot_aspp) 256]) = 304 channels: carried as 320 (zero channels behind the
# 304; the 304-channel weight of final1[0] is zero-padded in its packs, the state_dict keeps the reference shape) so that its
# K dimension is whole 128-byte tiles -- aligned / row-reuse convolution kernels instead of the per-thread tap tracking of the
# unaligned ones.  1 = the reference's 304.
# HRFP_LAZY (off = the reference's behaviour, deepv3.py:320-327: the HRFP branch runs in every forward): run only the part of the
# branch whose output the step reads -- none of it when neither O1 nor O2 is drawn, its first four layers when only O2 is.  Loss,
# gradients and every trainable tensor are unchanged; the BatchNorm running statistics of the (frozen, re-randomised) OC* layers
# -- buffers no output of the network depends on, in train or eval mode -- are then only updated by the passes that run.
cfg.MODEL.HRFP_LAZY = __import__("os").environ.get("MRFP_HRFP_LAZY", "0") == "1"
cfg.MODEL.DECODER_PAD = int(__import__("os").environ.get("MRFP_DECODER_PAD", "64"))
# BatchNorm statistics summed over all ranks of the default process group (reference config.py:92-93, torch.nn.SyncBatchNorm).
# Off: with 16 images per GPU the per-replica statistics ARE the reference's single-GPU population.
cfg.MODEL.SYNC_BN = False
# directory searched for ImageNet checkpoints when pretrained=True (no network access here)
cfg.MODEL.PRETRAINED_DIR = None


def assert_and_infer_cfg(args=None, make_immutable=True, train_mode=True):
    """reference config.py:95-128.  `args.syncbn` turns cfg.MODEL.SYNC_BN on (cross-rank statistics in the HIP BatchNorm, what
    torch.nn.SyncBatchNorm does in the reference); without it the statistics are per replica -- the reference's single-GPU
    semantics."""
    if args is not None and getattr(args, "syncbn", False):
        cfg.MODEL.SYNC_BN = True
    if make_immutable:
        cfg.immutable(True)
