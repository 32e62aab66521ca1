"""SPH3D_s3dis call pattern on s3g_util (torch restatement of models/SPH3D_s3dis.py:11-133).

This is harness code for bench.py / smoke(): it exists so that the headline metric
(point-cloud blocks/s, forward+backward, SPH3D_s3dis, 8192-point blocks) exercises the ops through
the same sequence of s3g_util calls, with the same shapes, as the reference's model graph.
The config mirrors s3dis_seg/s3dis_config.py:3-26.
"""
import copy
import os
import types

import torch
import torch.nn.functional as F

from .. import _lib
from .. import sph3gcn_util as s3g_util


def s3dis_config(num_input=8192):
    c = types.SimpleNamespace()
    c.num_input = num_input
    c.num_cls = 13
    c.mlp = 64
    c.num_sample = [2048, 768, 384, 128]
    c.radius = [0.1, 0.2, 0.4, 0.8]
    c.nn_uplimit = [64, 64, 64, 64]
    c.channels = [[128, 128], [256, 256], [256, 256], [512, 512]]
    c.multiplier = [[2, 2], [2, 2], [2, 2], [2, 2]]
    c.weight_decay = None
    c.kernel = [8, 2, 2]
    c.binSize = 8 * 2 * 2 + 1
    c.normalize = True
    c.pool_method = 'max'
    c.unpool_method = 'mean'
    c.sample = 'FPS'
    c.with_bn = True
    c.with_bias = False
    return c


def scannet_config(num_input=65536):
    """scannet_seg/scannet_config.py is the S3DIS plan with 21 classes at 8192 points; BASELINE config 5 asks for 65 536-point
    blocks: the sample counts scale with the input (x8)"""
    c = s3dis_config(num_input)
    c.num_cls = 21
    c.num_sample = [num_input // 4, num_input * 3 // 32, num_input * 3 // 64, num_input // 64]
    return c


def small_config(num_input=1024):
    """Reduced plan for smoke / CPU-oracle plumbing (BASELINE config #1 flavour)."""
    c = s3dis_config(num_input)
    c.num_sample = [256, 64]
    c.radius = [0.1, 0.2]
    c.nn_uplimit = [32, 32]
    c.channels = [[32, 32], [64, 64]]
    c.multiplier = [[2, 2], [2, 1]]
    c.mlp = 16
    return c


def normalize_xyz(points):
    """models/SPH3D_s3dis.py:11-19"""
    min_xyz, max_xyz = torch.aminmax(points, dim=1, keepdim=True)      # one reduction kernel instead of two
    center = (max_xyz + min_xyz) / 2
    xy = points[:, :, 0:2] - center[:, :, 0:2]
    z = points[:, :, 2:]
    return torch.cat((xy, z), dim=2)


def _net_input(points, config):
    xyz = points[:, :, 0:3]
    norm_xyz = normalize_xyz(xyz) if config.normalize else xyz
    return torch.cat((norm_xyz, points[:, :, 6:]), dim=2)


def _separable_conv3d_block(net, list_channels, bin_size, nn_index, nn_count, filt_idx, name,
                            depth_multiplier=None, weight_decay=None, reuse=None, with_bn=True,
                            with_bias=True, is_training=None):
    """models/SPH3D_s3dis.py:22-32"""
    for l, num_out_channels in enumerate(list_channels):
        scope = name + '_' + str(l + 1)
        net = s3g_util.separable_conv3d(net, num_out_channels, bin_size, depth_multiplier[l], scope, nn_index,
                                        nn_count, filt_idx, weight_decay=weight_decay, with_bn=with_bn,
                                        with_bias=with_bias, reuse=reuse, is_training=is_training)
    return net


_side_stream = {}
# sampling streams the plans rotate over (set before the first plan of a device).  1: every plan's sampling chain runs behind the
# previous one's — right when a step has more feature-path work than sampling (the S3DIS training step: 9 ms vs 3 ms; with 2
# the chains of consecutive steps overlap and take CUs from the feature path: 1763 vs 1787 blocks/s).  More when the chain is
# what a step waits for: the forward-only loop (2: 3.51 -> 3.33 ms) and the 65 536-point plan (2: 53.4 -> 29.7 ms).
SAMPLING_STREAMS = 1
_USE_ARENA = True          # one allocation per plan and producing stream (GraphPlan)
FUSE_POOL_GRAPH = os.environ.get("SPH3D_FUSE_POOL_GRAPH", "1") != "0"     # pooling rows + counts + transpose count pass in one launch
_ARENA_NEED = {}           # (points shape, config, ...) -> bytes the sampling / graph stream's tensors of such a plan took


class GraphPlan:
    """All graph-construction ops of one forward (they depend on xyz only).  Same ops, arguments and
    results as the build_graph / build_graph_deconv / spherical_kernel calls of models/SPH3D_s3dis.py:53-98; only
    the ISSUE ORDER and the STREAMS differ.  On the GPU three HIP streams run concurrently:
      * sampling stream: the FPS chain — m strictly sequential rounds on one workgroup per cloud (16 of 256 CUs);
      * graph stream   : every neighbour search, kernel binning, pooled-row gather and (for the backward pass) the
                         transposed graphs, level by level — VALU / integer-atomic bound work;
      * main stream    : the feature path (convolutions, GEMMs, normalisation) — L2 / MFMA / HBM bound work,
    and the main stream waits, per level, only on the event of the graph it is about to use.
    On CPU tensors (oracle-backed tests) everything is built lazily on the spot."""

    def __init__(self, points, config, overlap=True, points_ready=None, decoder=True, global_kernel=None, global_radius=100.0,
                 global_query=None, prepare_input=True, need_backward=None, xyz_transform=None):
        """decoder=False: an encoder-only plan (the classification net); global_kernel: also the global graph of
        models/SPH3D_modelnet.py:83-93 (query = centroid of the last level's points, every remaining point a neighbour) with
        the bins of that kernel (global_query: the query points [B, 1, 3], default the centroid of the last level);
        need_backward (default: torch.is_grad_enabled()): also build the transposed graphs the gradients gather over;
        prepare_input=False: `points` are coordinates only (no S3DIS input features to prepare).
        xyz_transform: a function of the raw coordinates (the classification net's unit-sphere normalisation) applied ON THE
        SAMPLING STREAM before anything else: the plan is built on its result (xyz_layers[0]; `xyz0()` hands it to the feature
        path), so that with points_ready the whole plan depends on the input batch alone.  global_query="centroid": the
        global graph's query is the mean of those coordinates (models/SPH3D_modelnet.py:44), computed there as well.
        points_ready: optional event after which `points` is valid.  With it the two side streams wait only for the
        INPUT, not for everything queued on the main stream — so when the host runs ahead (it issues a step in about
        half the time the GPU needs), the sampling / graph construction of step t+1 overlaps the backward pass of step
        t instead of idling the main stream at every step boundary (measured: 5.8 ms of main-stream idle per step)."""
        self.config = config
        self.need_backward = torch.is_grad_enabled() if need_backward is None else bool(need_backward)
        self.decoder = bool(decoder)
        self.global_kernel, self.global_radius, self.global_query = global_kernel, float(global_radius), global_query
        self._glob, self._glob_ev = None, None
        xyz = points[:, :, 0:3]
        self.use_side = bool(overlap and xyz.is_cuda)
        self.xyz_layers, self.indices, self.events = [xyz], [], []
        self._enc, self._dec = {}, {}
        self._enc_ev, self._dec_ev, self._pool_ev = {}, {}, {}
        self._synced = set()
        self.net_input = None
        self._in_key = None
        if self.use_side:
            self.main = torch.cuda.current_stream()
            streams = _side_stream.get(xyz.device)
            if streams is None:
                # (stream priorities were measured, round 2: side streams at the lowest and the feature path at the highest
                # priority change nothing — 11.66 vs 11.68 ms per step; queue priority does not stop resident workgroups
                # of the three streams from sharing the CUs)
                streams = _side_stream[xyz.device] = ([torch.cuda.Stream(device=xyz.device) for _ in range(SAMPLING_STREAMS)],
                                                      torch.cuda.Stream(device=xyz.device), [0])
            # the sampling chain is a few thousand strictly sequential rounds on 16 CUs: consecutive plans (whose inputs are
            # ready) take alternate sampling streams, so that the chain of the next batch runs beside this one's instead of behind
            # it — what bounds a forward-only loop (3.2 ms of sampling per 3.5-ms forward) and the 65 536-point plan
            s_fps = streams[0][streams[2][0] % len(streams[0])]
            streams[2][0] += 1
            s_graph = streams[1]
            # (Measured round 2, A/B in one gpurun call: the level-0 search kernel — one 1024-thread, 112-KB-LDS workgroup
            # per CU — cannot share a CU with a resident FPS workgroup (16 + 16 waves at 72 VGPRs), so beside the FPS chain
            # its last 16 workgroups run as a second round: 1.15 ms instead of 0.62 (tools/exp_graph_fps.py).  Running the
            # search IN FRONT of the chain — on the graph stream with an event, or on the sampling stream itself — made the
            # step slower, 1440 / 1457 vs 1478 blocks/s: the 2.9-ms sampling chain is what the deeper graph levels wait for,
            # and 0.6 ms more in front of it costs more than the search's second round.  Capping both kernels at 64 VGPRs so
            # that they do share a CU: 1467 vs 1475.  Left as it is.)
            if points_ready is not None:
                s_fps.wait_event(points_ready)
                s_graph.wait_event(points_ready)
            else:
                s_fps.wait_stream(self.main)
                s_graph.wait_stream(self.main)
            # tensors made on the main stream and read by the side streams' kernels (a temporary such as the classification
            # net's normalised coordinates may be freed by the caller while those kernels are still queued)
            points.record_stream(s_fps)
            points.record_stream(s_graph)
            if torch.is_tensor(global_query):
                global_query.record_stream(s_graph)
            # one block per producing stream for the plan's ~110 index / graph tensors (_lib.Arena: the allocator records one
            # event per block and consuming stream when a block is freed — a burst of ~100 marker packets in the main
            # stream's queue per step, 0.47 ms without a kernel); sized by the previous plan of the same shapes
            akey = (tuple(points.shape), id(config), bool(decoder), self.need_backward)
            need = _ARENA_NEED.get(akey, (0, 0))
            with torch.cuda.stream(s_fps), _lib.arena_scope(_lib.Arena(need[0], xyz.device) if _USE_ARENA else None) as a_fps:
                # one contiguous copy of the coordinates for every op of the plan (the [:, :, 0:3] view made each
                # neighbour search / binning / sampling call copy it again)
                xyz = xyz.contiguous()
                if xyz_transform is not None:
                    xyz = xyz_transform(xyz).contiguous()
                if isinstance(self.global_query, str):           # "centroid"
                    self.global_query = xyz.mean(dim=1, keepdim=True)
                    self.global_query.record_stream(s_graph)
                self.xyz_layers[0] = xyz
                ev_xyz = torch.cuda.Event()
                ev_xyz.record(s_fps)
                self._xyz_ev = ev_xyz
                # the network's input features (centred coordinates + colours, models/SPH3D_s3dis.py:11-19,38-41) depend on the
                # batch only: prepared here, ahead of the feature path (one reduction + four small kernels, 60 us of main-stream time)
                if prepare_input:
                    self.net_input = _net_input(points, config)
                    self._in_key = (points.data_ptr(), points._version, tuple(points.shape))
                    self._in_ev = torch.cuda.Event()
                    self._in_ev.record(s_fps)
                self._sampling_chain(s_fps)
            s_graph.wait_event(ev_xyz)
            # every tensor the sampling stream allocated is read by kernels on the graph stream (neighbour search,
            # binning, pooled-row gathers): tell the caching allocator, or their blocks return to the sampling stream's
            # pool when the plan is dropped and the NEXT step's sampling (which only waits for its input batch) may
            # overwrite them while this step's graph kernels still read them
            for t in self.xyz_layers + self.indices:
                if torch.is_tensor(t):
                    t.record_stream(s_graph)
            with torch.cuda.stream(s_graph), _lib.arena_scope(_lib.Arena(need[1], xyz.device) if _USE_ARENA else None) as a_graph:
                self._build_all(s_graph)
            if _USE_ARENA:
                # (+ 1/16: a plan on other data of the same shape may ask for a little more, e.g. another count of active bins)
                # (whole 32-MB units: plans of nearby sizes then ask the caching allocator for the same block size)
                unit = lambda n: ((n + (n >> 4) + (32 << 20) - 1) >> 25) << 25 if n > (8 << 20) else n + (n >> 4)
                want = (unit(a_fps.need), unit(a_graph.need))
                if want[0] > need[0] or want[1] > need[1]:
                    _ARENA_NEED[akey] = (max(want[0], need[0]), max(want[1], need[1]))
                    while len(_ARENA_NEED) > 16:
                        _ARENA_NEED.pop(next(iter(_ARENA_NEED)))
        else:
            if xyz_transform is not None:
                self.xyz_layers[0] = xyz_transform(xyz)
            if isinstance(self.global_query, str):
                self.global_query = self.xyz_layers[0].mean(dim=1, keepdim=True)
            self._sampling_chain(None)

    def xyz0(self):
        """the coordinates the plan was built on (after xyz_transform), ordered before the current stream's later work"""
        if self.use_side:
            self._sync(("xyz0",), self._xyz_ev, [self.xyz_layers[0]] + ([self.global_query] if torch.is_tensor(self.global_query) else []))
        return self.xyz_layers[0]

    def _sampling_chain(self, side):
        """FPS level after level (each level samples the previous level's samples)."""
        config = self.config
        for l in range(len(config.radius)):
            if config.num_sample[l] > 1:
                cur = self.xyz_layers[-1]
                if config.sample == 'FPS':
                    sample_index = s3g_util.farthest_point_sample(config.num_sample[l], cur)
                elif config.sample == 'random':
                    sample_index = s3g_util.random_sample(config.num_sample[l], cur)
                else:
                    raise ValueError('Unknown sampling method.')
                B = cur.shape[0]
                batch_indices = torch.arange(B, dtype=torch.int32, device=cur.device).view(-1, 1, 1)
                indices = torch.cat([batch_indices.expand(B, config.num_sample[l], 1),
                                     sample_index.unsqueeze(2).to(torch.int32)], dim=2)      # util.py:43-45
                self.indices.append(indices)
                self.xyz_layers.append(s3g_util.gather_nd(cur, indices))
            else:
                self.indices.append(None)
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(side)
                self.events.append(ev)

    # ---- graph construction proper (called on the graph stream, or lazily on CPU) ----
    def _make_enc(self, l):
        c = self.config
        xyz = self.xyz_layers[l]
        # util.py:29 + models/SPH3D_s3dis.py:57: neighbour graph and bins of one point set, one fused kernel on the GPU
        idx, cnt, dst, filt = s3g_util.build_intra_graph(xyz, c.radius[l], c.nn_uplimit[l], c.kernel, **self._bw_kw())
        return dict(intra_idx=idx, intra_cnt=cnt, filt_idx=filt)

    def _bw_kw(self):
        # (positional compatibility with the oracle-backed stand-ins of the CPU tests: the keyword only when it is not the default)
        return {} if self.need_backward else {"with_transpose": False}

    def _make_pool(self, l, g):
        fused = getattr(s3g_util, "gather_pooling_graph", None)        # (the oracle-backed CPU stand-ins of the tests have none)
        if fused is not None and g["intra_idx"].is_cuda and FUSE_POOL_GRAPH:
            # both gathers + (max pooling, training) the counting pass of the pooling graph's transpose in one launch
            g["inter_idx"], g["inter_cnt"] = fused(g["intra_idx"], g["intra_cnt"], self.indices[l],
                                                   with_transpose=self.config.pool_method == 'max' and self.need_backward)
            return
        g["inter_idx"] = s3g_util.gather_nd(g["intra_idx"], self.indices[l])       # models/SPH3D_s3dis.py:68-72
        g["inter_cnt"] = s3g_util.gather_nd(g["intra_cnt"], self.indices[l])
        if self.config.pool_method == 'max' and g["inter_idx"].is_cuda and self.need_backward:
            # the max-pool gradient gathers over the transposed pooling graph when one exists (tf_pool3d): built here, off the
            # critical path, like the transposes of the convolution graphs
            from .. import _tgraph
            # (rows gathered from the ball query's rows: ascending, distinct neighbour ids)
            _tgraph.transpose(g["inter_idx"], g["inter_cnt"], self.xyz_layers[l].shape[1], unique_rows=True)

    def _make_dec(self, l):
        c = self.config
        L = len(c.radius)
        radius, uplimit = c.radius[L - 1 - l], c.nn_uplimit[L - 1 - l]
        xyz_c, xyz_unpool = self.xyz_layers[L - l], self.xyz_layers[L - 1 - l]     # = reversed(xyz_layers)[l], [l + 1]
        # build_graph_deconv (util.py:52-58) + spherical_kernel: the intra half fused, the inter search as is
        intra_idx, intra_cnt, intra_dst, filt_idx = s3g_util.build_intra_graph(xyz_c, radius, uplimit, c.kernel, **self._bw_kw())
        from .. import tf_nnquery
        if (self.need_backward and xyz_c.is_cuda and c.unpool_method == 'mean' and s3g_util.neighbor_fn is s3g_util.build_sphere_neighbor
                and tf_nnquery.get_radius_mode() == "compat"):
            # the search also counts the in-edges: first pass of the transposed graph of the un-pooling gradient
            inter_idx, inter_cnt, inter_dst = tf_nnquery.build_sphere_neighbor_counted(xyz_c, xyz_unpool, radius, uplimit)
        else:
            inter_idx, inter_cnt, inter_dst = s3g_util.neighbor_fn(xyz_c, xyz_unpool, radius=radius, nnsample=uplimit)
        return dict(intra_idx=intra_idx, intra_cnt=intra_cnt, filt_idx=filt_idx, inter_idx=inter_idx,
                    inter_cnt=inter_cnt, inter_dst=inter_dst)

    def _pretranspose(self, g, n_src_conv, n_src_unpool=None):
        """build the transposed graphs the backward pass will ask for (cached in _tgraph), off the critical path"""
        if not self.need_backward:
            return
        from .. import _tgraph
        _tgraph.transpose(g["intra_idx"], g["intra_cnt"], n_src_conv, bin_index=g["filt_idx"],
                          num_bins=self.config.binSize)
        if n_src_unpool is not None and self.config.unpool_method == 'mean':
            _tgraph.transpose(g["inter_idx"], g["inter_cnt"], n_src_unpool)

    def _build_all(self, stream):
        """Graph stream schedule: every graph is issued as soon as the sampling levels it needs exist.  Decoder level l
        works on point sets L-l and L-1-l, so the LARGE decoder graphs (the last decoder levels) depend only on the
        FIRST sampling levels and are built early, interleaved with the encoder levels; when the main stream reaches
        the decoder only the smallest graph is still outstanding."""
        c = self.config
        L = len(c.radius)

        def mark(table, key):
            ev = torch.cuda.Event()
            ev.record(stream)
            table[key] = ev

        def make_dec(l):
            if not self.decoder:
                return
            g = self._make_dec(l)
            n_c = self.xyz_layers[L - l].shape[1]
            self._pretranspose(g, n_c, n_src_unpool=n_c)
            self._dec[l] = g
            mark(self._dec_ev, l)

        g = self._make_enc(0)
        self._pretranspose(g, self.xyz_layers[0].shape[1])
        self._enc[0] = g
        mark(self._enc_ev, 0)
        for s in range(L):                       # s = sampling level that has just become available
            stream.wait_event(self.events[s])
            if c.num_sample[s] > 1:
                self._make_pool(s, self._enc[s])
                mark(self._pool_ev, s)
            if s + 1 < L:
                g = self._make_enc(s + 1)
                self._pretranspose(g, self.xyz_layers[s + 1].shape[1])
                self._enc[s + 1] = g
                mark(self._enc_ev, s + 1)
            make_dec(L - 1 - s)                  # needs point sets s+1 and s
        if self.global_kernel is not None:
            self._glob = self._make_global()
            self._glob_ev = torch.cuda.Event()
            self._glob_ev.record(stream)

    def _make_global(self):
        """models/SPH3D_modelnet.py:83-93: one query per cloud (the centroid of the level's points), every point a neighbour"""
        xyz = self.xyz_layers[-1]
        query = self.global_query if self.global_query is not None else xyz.mean(dim=1, keepdim=True)
        nn_idx, nn_cnt, nn_dst = s3g_util.build_global_graph(xyz, query, self.global_radius)
        filt = s3g_util.spherical_kernel(xyz, query, nn_idx, nn_cnt, nn_dst, self.global_radius, kernel=self.global_kernel)
        return dict(nn_idx=nn_idx, nn_cnt=nn_cnt, filt_idx=filt, query=query)

    def glob(self):
        if self.use_side:
            self._sync(("glob",), self._glob_ev, list(self._glob.values()) + [self.xyz_layers[-1]])
        elif self._glob is None:
            self._glob = self._make_global()
        return self._glob

    def _sync(self, key, ev, tensors):
        if key in self._synced:
            return
        self.main.wait_event(ev)
        for t in tensors:
            if torch.is_tensor(t):
                t.record_stream(self.main)
        self._synced.add(key)

    def input(self, points):
        """centred coordinates + colours of the batch (prepared on the sampling stream when the plan runs on side streams)"""
        # the cached tensor belongs to the batch the plan was built from: a plan reused with other features on the same
        # coordinates (same xyz, new colours) gets them recomputed
        if (self.use_side and self.net_input is not None
                and self._in_key == (points.data_ptr(), points._version, tuple(points.shape))):
            self._sync(("input",), self._in_ev, [self.net_input])
            return self.net_input
        return _net_input(points, self.config)

    def enc(self, l):
        """encoder level l: intra graph + bins of xyz_l"""
        if self.use_side:
            self._sync(("enc", l), self._enc_ev[l], [self._enc[l]["intra_idx"], self._enc[l]["intra_cnt"],
                                                     self._enc[l]["filt_idx"], self.xyz_layers[l]])
        elif l not in self._enc:
            self._enc[l] = self._make_enc(l)
        return self._enc[l]

    def pool(self, l):
        """rows of the level-l intra graph at the sampled points"""
        g = self.enc(l)
        if self.use_side:
            self._sync(("pool", l), self._pool_ev[l], [g["inter_idx"], g["inter_cnt"]] + self.xyz_layers[l + 1:l + 2])
        elif "inter_idx" not in g:
            self._make_pool(l, g)
        return g

    def dec(self, l):
        if self.use_side:
            self._sync(("dec", l), self._dec_ev[l], list(self._dec[l].values()))
        elif l not in self._dec:
            self._dec[l] = self._make_dec(l)
        return self._dec[l]


def build_graphs(points, config, overlap=True):
    """Eagerly build every graph of one forward -> GraphPlan (kept for tests / smoke that inspect the graphs)."""
    plan = GraphPlan(points, config, overlap=overlap)
    for l in range(len(config.radius)):
        plan.enc(l)
        if config.num_sample[l] > 1:
            plan.pool(l)
    for l in range(len(config.radius)):
        plan.dec(l)
    return plan


def get_model(points, is_training, config=None, graphs=None, points_ready=None):
    """models/SPH3D_s3dis.py:35-113 (config lists are not reversed in place here)."""
    end_points = {}
    reuse = None
    plan = graphs if graphs is not None else GraphPlan(points, config, points_ready=points_ready)
    net = plan.input(points)
    net = s3g_util.pointwise_conv3d(net, config.mlp, 'mlp1', weight_decay=config.weight_decay,
                                    with_bn=config.with_bn, with_bias=config.with_bias, reuse=reuse,
                                    is_training=is_training)
    encoder = []
    for l in range(len(config.radius)):
        g = plan.enc(l)
        net = _separable_conv3d_block(net, config.channels[l], config.binSize, g["intra_idx"], g["intra_cnt"],
                                      g["filt_idx"], 'conv' + str(l + 1), config.multiplier[l], reuse=reuse,
                                      weight_decay=config.weight_decay, with_bn=config.with_bn,
                                      with_bias=config.with_bias, is_training=is_training)
        if config.num_sample[l] > 1:
            g = plan.pool(l)
            # the level's features go to the pooling AND, as the skip connection, to the decoder (models/SPH3D_s3dis.py:60-72)
            net, skip = s3g_util.pool3d_with_skip(net, g["inter_idx"], g["inter_cnt"], method=config.pool_method,
                                                  scope='pool' + str(l + 1))
            encoder.append(skip)
        else:
            encoder.append(net)
    channels = list(reversed(config.channels))
    multiplier = list(reversed(config.multiplier))
    encoder.reverse()
    for l in range(len(channels)):
        g = plan.dec(l)
        net = _separable_conv3d_block(net, channels[l], config.binSize, g["intra_idx"], g["intra_cnt"],
                                      g["filt_idx"], 'deconv' + str(l + 1), multiplier[l], reuse=reuse,
                                      weight_decay=config.weight_decay, with_bn=config.with_bn,
                                      with_bias=config.with_bias, is_training=is_training)
        net = s3g_util.unpool3d(net, g["inter_idx"], g["inter_cnt"], g["inter_dst"], method=config.unpool_method,
                                scope='unpool' + str(l + 1))
        if l + 1 < len(channels):
            # tf.concat((net, encoder[l]), axis=2) (models/SPH3D_s3dis.py:100-104) feeds the next level's separable convolution,
            # which takes the pair as it is (s3g_util.separable_conv3d: the depthwise kernels read both tensors in place)
            net = (net, encoder[l])
    # the last concatenation (models/SPH3D_s3dis.py:104) feeds only the logits layer: that layer reads its two halves in place
    # (s3g_util.pointwise_conv3d_concat); end_points['feats'] materialises the concatenation when somebody asks for it
    end_points = _EndPoints(end_points)
    end_points.feats_parts = (net, encoder[len(channels) - 1])
    net = s3g_util.pointwise_conv3d_concat(net, encoder[len(channels) - 1], config.num_cls, scope='logits', with_bn=False,
                                           with_bias=config.with_bias, activation_fn=None, is_training=is_training)
    return net, end_points


class _EndPoints(dict):
    """end_points whose 'feats' entry (the concatenated decoder output) is built on first access"""
    feats_parts = None

    def __missing__(self, key):
        if key == 'feats' and self.feats_parts is not None:
            self['feats'] = torch.cat(self.feats_parts, dim=2)
            return self['feats']
        raise KeyError(key)


class _MaskedXentFn(torch.autograd.Function):
    """the loss below as one kernel that also produces the gradient with respect to the logits (include/sph3d.h:
    sph3d_masked_softmax_xent): -> [B, S], a block's loss in S shares"""

    @staticmethod
    def forward(ctx, pred, label, inner_label):
        from .. import _lib
        B, N, C = pred.shape
        pred = _lib.f32(pred)
        label = label.reshape(B, N).long().contiguous()
        inner = inner_label.reshape(B, N).float().contiguous()
        S = _lib.lib().sph3d_masked_softmax_xent_parts(N)
        loss_part = torch.empty((B, S), dtype=torch.float32, device=pred.device)       # a block's loss in S slices of its points
        # (evaluation / no_grad: the losses only — no gradient pass, no [B, N, C] store)
        dlogits = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        _lib.check(_lib.lib().sph3d_masked_softmax_xent(B, N, C, _lib.ptr(pred), _lib.ptr(label), _lib.ptr(inner),
                                                        _lib.ptr(loss_part), _lib.ptr(dlogits), _lib.stream_ptr()))
        if dlogits is not None:
            ctx.save_for_backward(dlogits)
        return loss_part

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g[:, 0].reshape(-1, 1, 1), None, None       # (the slices of a block share its upstream gradient)


def get_loss(pred, label, end_points, inner_label):
    """models/SPH3D_s3dis.py:116-133: sum over the batch of the mean cross-entropy over inner points."""
    B, N, C = pred.shape
    if pred.is_cuda:
        return _MaskedXentFn.apply(pred, label, inner_label).sum()
    loss = F.cross_entropy(pred.reshape(-1, C), label.reshape(-1), reduction='none').reshape(B, N)
    mask = (inner_label > 0).to(loss.dtype)
    cnt = mask.sum(dim=1)
    per_block = (loss * mask).sum(dim=1) / cnt.clamp(min=1.0)
    per_block = torch.where(cnt > 0, per_block, torch.zeros_like(per_block))
    return per_block.sum()


class SPH3DS3DIS(torch.nn.Module):
    """Holds the VariableStore so parameters register with the optimiser; forward = get_model."""

    def __init__(self, config=None, device=None, seed=7):
        super().__init__()
        self.config = copy.deepcopy(config) if config is not None else s3dis_config()
        self.store = s3g_util.VariableStore(device=device, seed=seed)

    def forward(self, points, is_training=True, graphs=None, points_ready=None):
        with s3g_util.variable_store(self.store):
            return get_model(points, is_training, self.config, graphs=graphs, points_ready=points_ready)

    def loss(self, pred, label, inner_label):
        return get_loss(pred, label, None, inner_label)
