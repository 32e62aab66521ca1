"""SPH3D_shapenet call pattern on s3g_util (torch restatement of models/SPH3D_shapenet.py:33-123).

Part segmentation plan (shapenet_seg/shapenet_config.py): 2048-point objects, raw xyz as the only input feature, encoder
2048 -> 1024 -> 768 -> 384 -> 128 with radii .08/.16/.32/.64, the S3DIS channel plan, a decoder that mirrors it, then
mlp2 (-> 64), concatenation with the mlp1 features and a point-wise classifier.  The graph construction is the same
three-stream GraphPlan as the S3DIS harness (same ops and arguments as the reference's build_graph /
build_graph_deconv / spherical_kernel calls, issued ahead of the feature path).
"""
import copy
import types

import torch
import torch.nn.functional as F

from .. import sph3gcn_util as s3g_util
from .s3dis_net import GraphPlan, _separable_conv3d_block


def shapenet_config(num_input=2048):
    """shapenet_seg/shapenet_config.py:3-25"""
    c = types.SimpleNamespace()
    c.num_input = num_input
    c.mlp = 64
    c.num_sample = [1024, 768, 384, 128]
    c.radius = [0.08, 0.16, 0.32, 0.64]
    c.nn_uplimit = [64, 64, 64, 64]
    c.channels = [[128, 128], [256, 256], [256, 256], [512, 512]]
    c.multiplier = [[2, 2], [2, 2], [2, 2], [2, 2]]
    c.weight_decay = None
    c.kernel = [8, 2, 2]
    c.binSize = 8 * 2 * 2 + 1
    c.normalize = False
    c.pool_method = 'max'
    c.unpool_method = 'mean'
    c.sample = 'FPS'
    c.with_bn = True
    c.with_bias = False
    return c


def small_config(num_input=512):
    c = shapenet_config(num_input)
    c.mlp = 16
    c.num_sample = [128, 32]
    c.radius = [0.16, 0.32]
    c.nn_uplimit = [32, 32]
    c.channels = [[32, 32], [64, 64]]
    c.multiplier = [[2, 2], [2, 1]]
    return c


def get_model(points, num_cls, is_training, config=None, graphs=None, points_ready=None):
    """models/SPH3D_shapenet.py:33-113 (config lists are not reversed in place here)."""
    end_points = {}
    reuse = None
    net = s3g_util.pointwise_conv3d(points, config.mlp, 'mlp1', weight_decay=config.weight_decay,
                                    with_bn=config.with_bn, with_bias=config.with_bias, reuse=reuse,
                                    is_training=is_training)
    plan = graphs if graphs is not None else GraphPlan(points, config, points_ready=points_ready)
    encoder = [net]
    for l in range(len(config.radius)):
        g = plan.enc(l)
        net = _separable_conv3d_block(net, config.channels[l], config.binSize, g["intra_idx"], g["intra_cnt"],
                                      g["filt_idx"], 'conv' + str(l + 1), config.multiplier[l], reuse=reuse,
                                      weight_decay=config.weight_decay, with_bn=config.with_bn,
                                      with_bias=config.with_bias, is_training=is_training)
        if config.num_sample[l] > 1:
            g = plan.pool(l)
            # the level's features go to the pooling and, as the skip connection, to the decoder (as in s3dis_net)
            net, skip = s3g_util.pool3d_with_skip(net, g["inter_idx"], g["inter_cnt"], method=config.pool_method,
                                                  scope='pool' + str(l + 1))
            encoder.append(skip)
        else:
            encoder.append(net)
    channels = list(reversed(config.channels))
    multiplier = list(reversed(config.multiplier))
    encoder.reverse()                      # [level L, ..., level 1, mlp1]
    for l in range(len(channels)):
        g = plan.dec(l)
        net = _separable_conv3d_block(net, channels[l], config.binSize, g["intra_idx"], g["intra_cnt"],
                                      g["filt_idx"], 'deconv' + str(l + 1), multiplier[l], reuse=reuse,
                                      weight_decay=config.weight_decay, with_bn=config.with_bn,
                                      with_bias=config.with_bias, is_training=is_training)
        net = s3g_util.unpool3d(net, g["inter_idx"], g["inter_cnt"], g["inter_dst"], method=config.unpool_method,
                                scope='unpool' + str(l + 1))
        # tf.concat((net, encoder[l]), axis=2): the next level's separable convolution takes the pair (s3dis_net); the last
        # concatenation feeds a pointwise layer and is materialised
        net = (net, encoder[l]) if l + 1 < len(channels) else torch.cat((net, encoder[l]), dim=2)
    net = s3g_util.pointwise_conv3d(net, config.mlp, 'mlp2', weight_decay=config.weight_decay,
                                    with_bn=config.with_bn, with_bias=config.with_bias, reuse=reuse,
                                    is_training=is_training)
    net = torch.cat((net, encoder[-1]), dim=2)
    end_points['feats'] = net
    net = s3g_util.pointwise_conv3d(net, num_cls, scope='logits', with_bn=False, with_bias=config.with_bias,
                                    activation_fn=None, is_training=is_training)
    return net, end_points


def get_loss(pred, label, end_points=None):
    """models/SPH3D_shapenet.py:116-123: mean cross-entropy over all points"""
    C = pred.shape[-1]
    return F.cross_entropy(pred.reshape(-1, C), label.long().reshape(-1))


class SPH3DShapeNet(torch.nn.Module):
    def __init__(self, num_cls=3, config=None, device=None, seed=7):
        super().__init__()
        self.num_cls = num_cls
        self.config = copy.deepcopy(config) if config is not None else shapenet_config()
        self.store = s3g_util.VariableStore(device=device, seed=seed)

    def forward(self, points, is_training=True, graphs=None, points_ready=None):
        with s3g_util.variable_store(self.store):
            return get_model(points, self.num_cls, is_training, self.config, graphs=graphs, points_ready=points_ready)

    def loss(self, pred, label):
        return get_loss(pred, label)
