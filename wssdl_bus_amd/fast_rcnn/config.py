"""Hot-path configuration constants.

Mirrors the keys of the reference's global ``cfg`` (code/lib/fast_rcnn/config.py:28-31)
that the detection hot path reads; values and line numbers are the reference's.
``cfg`` is a plain attribute dict, mutable like the original.
"""


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def has_key(self, k):
        return k in self


__C = AttrDict()
cfg = __C

__C.TRAIN = AttrDict()
__C.TRAIN.LEARNING_RATE = 0.0005                    # config.py:40
__C.TRAIN.WEIGHT_DECAY = 0.0005                     # config.py:46
__C.TRAIN.WS_IMS_PER_BATCH = 2                      # :49
__C.TRAIN.WS_LOSS_USE_ADAPTIVE_SCALE_FACTOR = True  # :51
__C.TRAIN.WS_LOSS_SCALE_FACTOR = 0.5                # :52
__C.TRAIN.WS_MAL_PCT = 0.2209                       # :60
__C.TRAIN.MAX_GT_PER_IMAGE = 20                     # :92
__C.TRAIN.SCALES = (600,)                           # :109
__C.TRAIN.MAX_SIZE = 1000                           # :112
# augmentation of the image path (utils/blob.py): the reference's default.  skimage.transform.rotate / resize are
# restated from scikit-image 0.14.2's published algorithm as device kernels (csrc/image.hip; the library itself is
# absent): that part of the DEFAULT input pipeline is PARITY UNPINNED -- checked against this repo's own restatement
# and a second scipy implementation, never against a scikit-image run (README.md)
__C.TRAIN.USE_ROTATION = True                       # :136
__C.TRAIN.ROTATION_MAX_ANGLE = 5                    # :137
__C.TRAIN.USE_CROPPING = True                       # :140
__C.TRAIN.CROPPING_MAX_MARGIN = 0.05                # :141
__C.TRAIN.USE_BRIGHTNESS_ADJUSTMENT = True          # :144
__C.TRAIN.BRIGHTNESS_ADJUSTMENT_MAX_DELTA = 0.2     # :145
__C.TRAIN.USE_CONTRAST_ADJUSTMENT = True            # :148
__C.TRAIN.CONTRAST_ADJUSTMENT_LOWER_FACTOR = 0.2    # :149
__C.TRAIN.CONTRAST_ADJUSTMENT_UPPER_FACTOR = 1.8    # :150
__C.TRAIN.IMS_PER_BATCH = 1                         # :115
__C.TRAIN.BATCH_SIZE = 128                          # :118
__C.TRAIN.FG_FRACTION = 0.25                        # :121
__C.TRAIN.FG_THRESH = 0.5                           # :124
__C.TRAIN.BG_THRESH_HI = 0.5                        # :128
__C.TRAIN.BG_THRESH_LO = 0.0                        # :130
__C.TRAIN.BBOX_INSIDE_WEIGHTS = (1.0, 1.0, 1.0, 1.0)   # :178
__C.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED = False   # :181 (read by proposal_target_layer_tf_bus.py:221-224)
__C.TRAIN.BBOX_NORMALIZE_MEANS = (0.0, 0.0, 0.0, 0.0)  # :182
__C.TRAIN.BBOX_NORMALIZE_STDS = (0.1, 0.1, 0.2, 0.2)   # :183
__C.TRAIN.RPN_POSITIVE_OVERLAP = 0.7                # :196
__C.TRAIN.RPN_NEGATIVE_OVERLAP = 0.3                # :198
__C.TRAIN.RPN_CLOBBER_POSITIVES = False             # :200
__C.TRAIN.RPN_FG_FRACTION = 0.5                     # :202
__C.TRAIN.RPN_BATCHSIZE = 256                       # :204
__C.TRAIN.RPN_NMS_THRESH = 0.7                      # :206
__C.TRAIN.RPN_PRE_NMS_TOP_N = 12000                 # :208
__C.TRAIN.RPN_POST_NMS_TOP_N = 2000                 # :210
__C.TRAIN.RPN_MIN_SIZE = 16                         # :212
__C.TRAIN.RPN_BBOX_INSIDE_WEIGHTS = (1.0, 1.0, 1.0, 1.0)   # :214
__C.TRAIN.RPN_POSITIVE_WEIGHT = -1.0                # :218

__C.TEST = AttrDict()
__C.TEST.NMS = 0.3                                  # :238
__C.TEST.CLS_AGNOSTIC_NMS = False                   # :241
__C.TEST.BBOX_REG = True                            # :248
# (addition) the post-detection step (test_bus.py:360-401) as one device op when the tensors are on the GPU
__C.TEST.FUSED_POST_DETECTIONS = True
__C.TEST.RPN_NMS_THRESH = 0.7                       # :257
__C.TEST.RPN_PRE_NMS_TOP_N = 6000                   # :259
__C.TEST.RPN_POST_NMS_TOP_N = 300                   # :262
__C.TEST.RPN_MIN_SIZE = 16                          # :265

__C.RNG_SEED = 3                                    # :290
__C.EPS = 1e-14                                     # :293
__C.USE_GPU_NMS = False                             # :321.  True: nms_wrapper.nms and the proposal layer apply the CUDA
                                                    #  kernel's rule (iou > (float)thresh, nms_kernel.cu:71) instead of
                                                    #  cpu_nms's ((double)iou >= thresh); both run the HIP kernel

# --- additions of this implementation (not in the reference) -----------------
# 'reference': anchor / RoI sub-sampling draws from numpy's global legacy RandomState
#              exactly like the reference (bit-identical labels, one host round trip);
# 'device'   : counter-based hash sampling on the GPU (no host sync; same distribution).
__C.SAMPLING_RNG = "reference"
__C.DEVICE_RNG_SEED = 3
# f2: feed the proposal layer with the raw rpn_cls_score and fuse reshape -> softmax -> reshape
# into its decode kernel (the 'rpn_cls_prob*' layers are then not materialised)
__C.FUSED_RPN_SOFTMAX = False
# RoI-pool bin rounding: 'cuda' (canonical, roi_pooling_op_gpu.cu.cc:51-58) or
# 'cpu' (roi_pooling_op.cc:167-170)
__C.ROI_POOL_ROUNDING = "cuda"
# training path: the autograd pair hands a 1-byte arg-max from RoiPool to RoiPoolGrad when the
# library supports the shape (include/wssdl_bus_hip.h); False = the reference's i32 layout
__C.ROI_POOL_COMPACT_ARGMAX = True
# how the compact pair's device-side error flags (a RoI whose bin windows the 1-byte code cannot describe;
# backward lists that do not fit) are consumed: 'deferred' = no read-back, the train step polls them
# before every optimiser step and raises (roi_pooling_op.poll_flags); 'eager' = RoiPoolFunction reads
# the forward flag after every call and re-runs an overflowing call on the i32 pair, which takes any RoI
__C.ROI_POOL_FLAG_CHECK = "deferred"
# True: the proposal layer hands out a fixed-shape blob [N * post_nms_topN, 5] (unused rows carry batch
# index -1) and nothing between the backbone and the loss copies to the host: the hot path can be
# captured in a hipGraph.  The per-RoI head then runs on the padded row count (batch-norm masked to the
# live rows).  False (default): the blob is compacted, which costs one read-back of N counts per step.
# RoI-pool forward of the training path on launches with many proposals per image (round 6): 'auto' = the library's
# rule (wssdl_roi_pool_forward_blocks_auto: R >= 2048 and R >= 1200 x images -- the alternating weak step, the
# reference's default 1 + 2 batch) -> block-maximum tables of the step's feature map + bin rows walked in (image, first
# window row) order, 1.6-1.8 x faster there, the same bits; True = wherever the shape is supported, False = never (the
# rows kernel).  The tables are 3.75 x the feature map, allocated per call.
__C.ROI_POOL_FWD_BLOCKS = 'auto'
# RoI-pool backward of the training path on launches with few images (<= 4) and many RoIs per image (>= 1000):
# 'auto' = the library's rule (wssdl_roi_pool_backward_split_segments: 4 segments there, the exact walk
# everywhere else), an int = that many segments, 0 = always the exact walk.  The split form is deterministic but
# associates each element's f32 sum differently from the reference (within ~1e-7 of it; north_star's tolerance
# for RoI pooling is 1e-5); the parity tests of the gradient run the exact walk.
__C.ROI_POOL_BWD_SPLIT = 'auto'
# RoI-pool backward of the training path on train-sized launches: the bin-owner form (round 5: every bin read once by
# the tile of its window's first cell, halos merged in a fixed order; 10-25 % faster than the exact walk,
# deterministic, within ~1e-6 of the reference's ordered sum).  'auto' = the library's rule
# (wssdl_roi_pool_backward_owner_plan_for: R >= 1536 RoIs, C % 128 == 0, N * C >= 2048 (image, channel) pairs, pooled
# size <= 8 x 8 -> owner plan 8; anything else -1 = the split form / exact walk), an int = that owner plan, -1 = never.
# Takes precedence over the split form.  Its halo scratch is N * tiles * 42 cells * C * 4 bytes per backward call
# (179 MB at 8 x 38 x 63 x 1024, 2.3 x bottom_diff; only the halo cells are touched).  train_bus prints the form the
# first backward of a run takes (BackwardPlan.variant).
__C.ROI_POOL_BWD_OWNER = 'auto'
# waves per tile stream of the bin-owner form: 'auto' = the library's rule (wssdl_roi_pool_backward_owner_segments: more
# than one on launches with few (image, channel) pairs, whose longest streams the whole chip would otherwise wait for),
# an int = that many, 1 = the plain owner form.  Same lists, deterministic, the owner form's tolerance.
__C.ROI_POOL_BWD_OWNER_SEGMENTS = 'auto'
# True: the backward is ALWAYS the exact walk -- the reference's f32 summation order (roi, ph, pw), bit for bit
# (roi_pooling_op_gpu.cu.cc:132-186) -- whatever the two keys above say.  The default training gradient is
# tolerance-parity (north_star: 1e-5; measured <= 1e-6 of the tensor's scale), not bit-parity; every forward output
# and every integer result is bit-exact in both settings.
__C.ROI_POOL_BWD_EXACT = False
# the first backward of each launch class (images x channels x form) prints the form it takes to stderr
__C.ROI_POOL_ANNOUNCE_BWD_FORM = True
__C.PADDED_ROIS = False
# a13: the four supervised loss terms and their gradients as one device op (csrc/loss.hip) when the
# layers are on the GPU; False = the chain of torch ops in fast_rcnn/train_bus.py
__C.FUSED_LOSS = True


def cfg_from_list(cfg_list):
    """Set config keys via list (e.g., from command line), config.py:392-412."""
    from ast import literal_eval
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        key_list = k.split(".")
        d = __C
        for subkey in key_list[:-1]:
            assert subkey in d
            d = d[subkey]
        subkey = key_list[-1]
        assert subkey in d
        try:
            value = literal_eval(v)
        except Exception:
            value = v
        assert type(value) == type(d[subkey]), "type {} does not match original type {}".format(
            type(value), type(d[subkey]))
        d[subkey] = value
