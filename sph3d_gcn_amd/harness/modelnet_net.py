"""SPH3D_modelnet call pattern on s3g_util (torch restatement of models/SPH3D_modelnet.py:11-119).

Classification plan (modelnet40_cls/modelnet_config.py): encoder levels with the raw xyz concatenated to the
features of every level (odd channel counts: 35, 67, 131), per-level global max-pools, one global spherical
convolution (query = the cloud centroid, radius 100, K = number of remaining points, kernel [8,2,1] -> 17 bins)
and three fully connected layers with dropout 0.5.  Harness code: it drives the ops of the hot path through the
same sequence of s3g_util calls, with the same shapes, as the reference's model graph.
"""
import copy
import types

import torch
import torch.nn.functional as F

from .. import sph3gcn_util as s3g_util
from .s3dis_net import GraphPlan, _separable_conv3d_block


def modelnet_config(num_input=10000):
    """modelnet40_cls/modelnet_config.py:3-36"""
    c = types.SimpleNamespace()
    c.num_input = num_input
    c.num_cls = 40
    c.mlp = 32
    c.num_sample = [num_input // 4 ** (i + 1) for i in range(10) if (num_input // 4 ** (i + 1)) > 100]
    c.radius = [0.1, 0.2, 0.4][:len(c.num_sample)]
    c.nn_uplimit = [64, 64, 64][:len(c.num_sample)]
    c.channels = [[64, 64], [64, 128], [128, 128]][:len(c.num_sample)]
    c.multiplier = [[2, 1], [1, 2], [1, 1]][:len(c.num_sample)]
    c.global_channels = 512
    c.global_multiplier = 2
    c.weight_decay = 1e-5
    c.kernel = [8, 2, 2]
    c.binSize = 8 * 2 * 2 + 1
    c.normalize = True
    c.pool_method = 'max'
    c.sample = 'FPS'
    c.use_raw = True
    c.with_bn = True
    c.with_bias = False
    return c


def small_config(num_input=1024):
    """BASELINE config #1 flavour: 1024 points, two encoder levels (samples [256, 64]), ModelNet channel plan."""
    c = modelnet_config(num_input)
    c.num_sample = [256, 64]
    c.radius = [0.1, 0.2]
    c.nn_uplimit = [32, 32]
    c.channels = [[64, 64], [64, 128]]
    c.multiplier = [[2, 1], [1, 2]]
    c.global_channels = 128
    return c


def normalize_xyz(points):
    """models/SPH3D_modelnet.py:11-17: centre on the mean, scale the farthest point to unit distance"""
    points = points - points.mean(dim=1, keepdim=True)
    scale = points.pow(2).sum(dim=-1, keepdim=True).max(dim=1, keepdim=True)[0].sqrt()
    return points / scale


def get_model(points, is_training, config=None, dropout_generator=None, points_ready=None):
    """models/SPH3D_modelnet.py:33-107: points [B, N, 3] -> logits [B, num_cls]
    points_ready: optional event after which `points` is valid (GraphPlan: the plan of this step then overlaps the previous
    step's backward pass instead of waiting for it)"""
    batch_size, num_point = points.shape[0], points.shape[1]
    end_points = {}
    assert num_point == config.num_input
    reuse = None
    # On the GPU the graph construction (same ops, arguments and results as the build_graph / spherical_kernel calls below) is
    # issued ahead of the feature path on the sampling and graph streams, like the segmentation nets' (s3dis_net.GraphPlan):
    # the normalisation of the coordinates and the global viewing point (:35-44) are its first kernels there
    plan = None
    if points.is_cuda and config.sample == 'FPS':
        plan = GraphPlan(points[:, :, 0:3], config, decoder=False, global_kernel=[8, 2, 1], global_query="centroid",
                         prepare_input=False, points_ready=points_ready,
                         xyz_transform=normalize_xyz if config.normalize else None)
        xyz = plan.xyz0()
        query = None
    else:
        if config.normalize:
            points = normalize_xyz(points)
        xyz = points
        query = xyz.mean(dim=1, keepdim=True)               # the global viewing point
    net = s3g_util.pointwise_conv3d(xyz, config.mlp, 'mlp1', weight_decay=config.weight_decay,
                                    with_bn=config.with_bn, with_bias=config.with_bias, reuse=reuse,
                                    is_training=is_training)
    global_feat = []
    for l in range(len(config.radius)):
        if config.use_raw:
            net = torch.cat([net, xyz], dim=-1)
        if plan is not None:
            g = plan.enc(l)
            intra_idx, intra_cnt, filt_idx = g["intra_idx"], g["intra_cnt"], g["filt_idx"]
            xyz = plan.xyz_layers[l]
        else:
            intra_idx, intra_cnt, intra_dst, indices = s3g_util.build_graph(
                xyz, config.radius[l], config.nn_uplimit[l], config.num_sample[l], sample_method=config.sample)
            filt_idx = s3g_util.spherical_kernel(xyz, xyz, intra_idx, intra_cnt, intra_dst, config.radius[l],
                                                 kernel=config.kernel)
        net = _separable_conv3d_block(net, config.channels[l], config.binSize, intra_idx, intra_cnt, filt_idx,
                                      'conv' + str(l + 1), config.multiplier[l], reuse=reuse,
                                      weight_decay=config.weight_decay, with_bn=config.with_bn,
                                      with_bias=config.with_bias, is_training=is_training)
        if config.num_sample[l] > 1:
            if plan is not None:
                g = plan.pool(l)
                inter_idx, inter_cnt = g["inter_idx"], g["inter_cnt"]
                xyz = plan.xyz_layers[l + 1]
            else:
                xyz = s3g_util.gather_nd(xyz, indices)
                inter_idx = s3g_util.gather_nd(intra_idx, indices)
                inter_cnt = s3g_util.gather_nd(intra_cnt, indices)
            net = s3g_util.pool3d(net, inter_idx, inter_cnt, method=config.pool_method, scope='pool' + str(l + 1))
        global_feat.append(net.max(dim=1, keepdim=True)[0])

    # global feature extraction in the final layer (:83-93)
    global_radius = 100.0
    if plan is not None:
        g = plan.glob()
        nn_idx, nn_cnt, filt_idx = g["nn_idx"], g["nn_cnt"], g["filt_idx"]
    else:
        nn_idx, nn_cnt, nn_dst = s3g_util.build_global_graph(xyz, query, global_radius)
        filt_idx = s3g_util.spherical_kernel(xyz, query, nn_idx, nn_cnt, nn_dst, global_radius, kernel=[8, 2, 1])
    net = s3g_util.separable_conv3d(net, config.global_channels, 17, config.global_multiplier, 'global_conv', nn_idx,
                                    nn_cnt, filt_idx, reuse=reuse, weight_decay=config.weight_decay,
                                    with_bn=config.with_bn, with_bias=config.with_bias, is_training=is_training)
    global_feat.append(net)
    net = torch.cat(global_feat, dim=2)

    # MLP on the global point cloud vector (:96-105)
    training = True if is_training is None else bool(is_training)
    net = net.reshape(batch_size, -1)
    net = s3g_util.fully_connected(net, 512, scope='fc1', weight_decay=config.weight_decay, with_bn=config.with_bn,
                                   with_bias=config.with_bias, is_training=is_training)
    net = _dropout(net, 0.5, training, dropout_generator)
    net = s3g_util.fully_connected(net, 256, scope='fc2', weight_decay=config.weight_decay, with_bn=config.with_bn,
                                   with_bias=config.with_bias, is_training=is_training)
    net = _dropout(net, 0.5, training, dropout_generator)
    net = s3g_util.fully_connected(net, config.num_cls, scope='logits', with_bn=False, with_bias=config.with_bias,
                                   activation_fn=None, is_training=is_training)
    return net, end_points


def _dropout(x, rate, training, generator):
    """tf.layers.dropout(rate): inverted dropout; a generator makes the mask reproducible across devices"""
    if not training or rate <= 0.0:
        return x
    if generator is None:
        return F.dropout(x, rate, training=True)
    keep = (torch.rand(x.shape, generator=generator) >= rate).to(device=x.device, dtype=x.dtype)
    return x * keep / (1.0 - rate)


def get_loss(pred, label, end_points=None):
    """models/SPH3D_modelnet.py:110-119: mean sparse softmax cross-entropy"""
    return F.cross_entropy(pred, label.long().reshape(-1))


class SPH3DModelNet(torch.nn.Module):
    """Holds the VariableStore so parameters register with the optimiser; forward = get_model."""

    def __init__(self, config=None, device=None, seed=7):
        super().__init__()
        self.config = copy.deepcopy(config) if config is not None else modelnet_config()
        self.store = s3g_util.VariableStore(device=device, seed=seed)

    def forward(self, points, is_training=True, dropout_generator=None, points_ready=None):
        with s3g_util.variable_store(self.store):
            return get_model(points, is_training, self.config, dropout_generator=dropout_generator, points_ready=points_ready)

    def loss(self, pred, label):
        """train_modelnet.py:162-164: classification loss + the weight-decay 'losses' collection + the BN regularisers
        scaled by weight_decay"""
        total = get_loss(pred, label)
        wd = self.store.collect_losses()
        if wd is not None:
            total = total + wd
        reg = self.store.regularization_loss()
        if reg is not None and self.config.weight_decay is not None:
            total = total + self.config.weight_decay * reg
        return total
