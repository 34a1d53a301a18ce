"""Eager PyTorch mirror of the reference's graph-layer wrappers for the hot path.

Reference: code/lib/networks/network.py:21-40 (the ``@layer`` decorator and ``feed``),
:196-281 (roi_pool, proposal_layer, anchor_target_layer[_joint], proposal_target_layer[_joint]),
:283-291 (reshape_layer), :398-404 (softmax).  Same method names, argument order, tuple
unwrapping and output dtypes; the TF1 static graph becomes eager calls that run the HIP
kernels on the current stream.  All tensors are NHWC like the reference's.
"""
import torch

from ..roi_pooling_layer import roi_pooling_op as roi_pool_op
from ..rpn_msr.anchor_target_layer_tf_bus import (anchor_target_layer as anchor_target_layer_py,
                                                  anchor_target_layer_joint as anchor_target_layer_joint_py,
                                                  anchor_target_layer_ws as anchor_target_layer_ws_py)
from ..rpn_msr.proposal_layer_tf_bus import (proposal_layer as proposal_layer_py,
                                             proposal_layer_from_score as proposal_layer_from_score_py)
from ..rpn_msr.proposal_target_layer_tf_bus import (proposal_target_layer as proposal_target_layer_py,
                                                    proposal_target_layer_joint as proposal_target_layer_joint_py)


def layer(op):
    """The chaining protocol of network.py:21-40 as a decorator: a layer method consumes whatever was
    fed last (one tensor, or a list when several were fed), its result is stored in ``self.layers``
    under ``name`` (auto-numbered from the method's name when the caller gives none) and becomes
    the fed input of the next call; the call returns the network so that calls chain."""
    import functools

    @functools.wraps(op)
    def chained(net, *args, **kwargs):
        if 'name' not in kwargs:
            kwargs['name'] = net.get_unique_name(op.__name__)
        fed = net.inputs
        if not fed:
            raise RuntimeError('No input variables found for layer %s.' % kwargs['name'])
        result = op(net, fed[0] if len(fed) == 1 else list(fed), *args, **kwargs)
        net._register(kwargs['name'], result)
        return net
    return chained


def _blob(out):
    """[-1, 5] view of a proposal blob that keeps its per-image row-count tags."""
    r = out.reshape(-1, 5)
    for tag in ("_wssdl_counts", "_wssdl_pitch", "_wssdl_counts_dev"):
        c = getattr(out, tag, None)
        if c is not None:
            setattr(r, tag, c)
    return r


def _first(x):
    # "only use the first input": a tuple-valued layer (e.g. 'roi-data') feeds its element 0
    return x[0] if isinstance(x, tuple) else x


class Network(object):
    """``layers``: name -> tensor (or tuple of tensors); ``inputs``: what the next layer call consumes."""

    def __init__(self, inputs=None):
        self.inputs = []
        self.layers = dict(inputs or {})

    def _lookup(self, key):
        if key not in self.layers:
            raise KeyError('Unknown layer name fed: %s' % key)
        return self.layers[key]

    def _register(self, name, result):
        self.layers[name] = result
        self.inputs = [result]

    def feed(self, *args):
        """network.py:54-66: names are resolved through ``layers``, anything else is fed as it is."""
        assert len(args) != 0
        self.inputs = [self._lookup(a) if isinstance(a, str) else a for a in args]
        return self

    def get_output(self, l):
        """network.py:68-74."""
        return self._lookup(l)

    def get_unique_name(self, prefix):
        """network.py:76-78: ``<prefix>_<k>``, k = 1 + the number of layers whose name starts with prefix."""
        taken = [n for n in self.layers if n.startswith(prefix)]
        return '%s_%d' % (prefix, len(taken) + 1)

    # ------------------------------------------------------------ hot path ---
    @layer
    def roi_pool(self, input, pooled_height, pooled_width, spatial_scale, name):
        """network.py:196-210: returns top_data only (element [0] of the op's outputs);
        differentiable w.r.t. the feature map."""
        data, rois = _first(input[0]), _first(input[1])
        return roi_pool_op.roi_pool_autograd(data, rois, pooled_height, pooled_width,
                                             spatial_scale, return_argmax=False)[0]

    @layer
    def proposal_layer(self, input, _feat_stride, anchor_scales, is_training, is_ws, name):
        """network.py:212-216: -> [-1, 5] float32."""
        with torch.no_grad():
            out = proposal_layer_py(_first(input[0]).detach(), input[1].detach(), input[2],
                                    is_training, is_ws, _feat_stride, anchor_scales)
        return _blob(out)

    @layer
    def proposal_layer_from_score(self, input, _feat_stride, anchor_scales, is_training, is_ws, name):
        """f2 variant of proposal_layer: input[0] is the raw rpn_cls_score."""
        with torch.no_grad():
            out = proposal_layer_from_score_py(_first(input[0]).detach(), input[1].detach(), input[2],
                                               is_training, is_ws, _feat_stride, anchor_scales)
        return _blob(out)

    @layer
    def anchor_target_layer(self, input, _feat_stride, anchor_scales, dataset, is_ws, name):
        """network.py:218-233: tf.cond on is_ws between the two py_funcs; labels cast to int32."""
        with torch.no_grad():
            if is_ws:
                o = anchor_target_layer_ws_py(_first(input[0]), input[1], input[2], input[3],
                                              input[4], _feat_stride, anchor_scales)
            else:
                o = anchor_target_layer_py(_first(input[0]), input[1], input[2], input[3],
                                           input[4], _feat_stride, anchor_scales, dataset)
        return o[0].to(torch.int32), o[1], o[2], o[3]

    @layer
    def anchor_target_layer_joint(self, input, _feat_stride, anchor_scales, dataset, is_training,
                                  name):
        """network.py:235-249."""
        with torch.no_grad():
            o = anchor_target_layer_joint_py(_first(input[0]), input[1], input[2], input[3],
                                             input[4], is_training, _feat_stride, anchor_scales,
                                             dataset)
        return o[0].to(torch.int32), o[1], o[2], o[3]

    @layer
    def proposal_target_layer(self, input, classes, is_training, is_ws, name):
        """network.py:251-265: (rois [-1,5], labels int32, targets, inside_w, outside_w)."""
        with torch.no_grad():
            o = proposal_target_layer_py(_first(input[0]), input[1], input[2], classes,
                                         is_training, is_ws)
        return o[0].reshape(-1, 5), o[1].to(torch.int32), o[2], o[3], o[4]

    @layer
    def proposal_target_layer_joint(self, input, classes, is_training, name):
        """network.py:267-281."""
        with torch.no_grad():
            o = proposal_target_layer_joint_py(_first(input[0]), input[1], input[2], classes,
                                               is_training)
        return o[0].reshape(-1, 5), o[1].to(torch.int32), o[2], o[3], o[4]

    @layer
    def reshape_layer(self, input, d, name):
        """network.py:283-291 on NHWC tensors.  For d=2: out[n, a*H+h, w, c] = in[n,h,w,c*A+a];
        for 'rpn_cls_prob_reshape' (d=2A) the inverse map."""
        n, h, w, c = input.shape
        x = input.permute(0, 3, 1, 2)
        if name == 'rpn_cls_prob_reshape':
            x = x.reshape(n, int(d), int(float(h) / float(d) * float(c)), w)
        else:
            x = x.reshape(n, int(d), int(float(h) * (float(c) / float(d))), w)
        return x.permute(0, 2, 3, 1)

    @layer
    def softmax(self, input, name):
        """network.py:398-404: over the last axis."""
        return torch.softmax(input, dim=-1)
