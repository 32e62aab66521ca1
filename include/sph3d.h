/* sph3d.h — C ABI of libsph3d (hand-written HIP kernels for gfx950 / MI355X).
 *
 * Drop-in boundary for the SPH3D-GCN tf_ops hot path.  Each entry point
 * replaces ONE launcher of the reference (cited per function) and takes
 * exactly what that launcher took — plain ints/floats by value and raw DEVICE
 * pointers — plus a HIP stream.  Differences from the reference launchers, all
 * deliberate:
 *   - every call returns an int status (0 = ok, <0 = SPH3D_E*); the reference
 *     returned void and never checked a launch;
 *   - every call is stream-ordered on `stream` (hipStream_t passed as void*);
 *     nothing synchronises the device (the reference's pool/unpool launchers
 *     called cudaDeviceSynchronize, tf_pool3d_gpu.cu:97,104,111,118);
 *   - outputs do NOT need to be pre-zeroed by the caller: each op fully
 *     defines its outputs (unused neighbour slots = 0, as the reference's
 *     cudaMemset in OpKernel::Compute left them, e.g. tf_nnquery.cpp:100-102);
 *   - all tensors are dense row-major float32 / int32, as in the reference.
 *
 * Notation: B batch, N database/input points, M query/output points,
 * K = nn_sample cap, C in-channels, r depth multiplier, F = n*p*q+1 bins.
 *
 * No torch / TF types appear here.  The Python ops in sph3d_gcn_amd/ bind this
 * file with ctypes; INTEGRATION.md shows the equivalent TF OpKernel binding.
 */
#ifndef SPH3D_H
#define SPH3D_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sph3d_stream_t;   /* hipStream_t */

enum {
    SPH3D_OK = 0,
    SPH3D_EINVAL = -1,      /* bad dimension / attribute (what OP_REQUIRES rejected) */
    SPH3D_EWORKSPACE = -2,  /* workspace too small */
    SPH3D_ELAUNCH = -3,     /* hipGetLastError() != hipSuccess after launch */
    SPH3D_EUNSUPPORTED = -4 /* shape outside what the kernels are built for */
};

/* Library identification. */
/* bumped whenever an exported symbol is removed or changes its signature (2: round 5 removed the sph3d_conv_plan* /
 * sph3d_depthwise_conv3d_lds* entries; additions alone do not bump it) */
#define SPH3D_ABI_VERSION 2
int sph3d_abi_version(void);                 /* == SPH3D_ABI_VERSION of the header the library was built from */
const char* sph3d_last_error(void);          /* thread-local text of the last non-OK status */
const char* sph3d_build_info(void);          /* "gfx950 hipcc <ver> ..." */

/* ---- nnquery ------------------------------------------------------------
 * replaces buildSphereNeighborLauncher (tf_ops/nnquery/tf_nnquery_gpu.cu:115-121;
 * kernel cal_nn_binidx :15-65; op BuildSphereNeighbor tf_nnquery.cpp:55-111).
 * database[B,N,3], query[B,M,3] -> nn_index[B,M,K] i32 (ascending database
 * index, first K win, unused = 0), nn_count[B,M] i32 in [1,K],
 * nn_dist[B,M,K] f32 = sqrt(euclidean distance) (sic, :47,:54).
 * Reference semantics reproduced bit-exactly, including the radius-growth
 * chain (:59): query (i,j) is searched with the radius left behind by the
 * previous query of reference-thread (i mod 32, j mod 1024).
 * Growth is bounded: after SPH3D_MAX_GROWTH_PASSES empty passes the query is
 * stored with nn_count = 0 (the reference would spin forever).
 * When no query of a call needs that growth, the radius of a query is a function of its position in the chain alone, and
 * every query is independent (csrc/nngrid.hip): the early positions — radius <= 3 * radius, the queries that would otherwise
 * scan the whole cloud for a handful of hits — are searched over a cell grid, the late ones by an early-stopping scan per
 * query.  Whenever some query does need the growth (device-side flag, no host round trip) the chain walk over the cloud
 * (csrc/nnquery.hip) computes the call.  Same rows bit for bit either way.
 * Where the grid lives: the entry points WITH the reference launcher's signature (no workspace argument:
 * sph3d_build_sphere_neighbor[_fixed], sph3d_build_sphere_graph[_ocml]) are conveniences that keep one library-owned
 * device buffer per (device, stream) — hipMalloc on first use, hipFree (a device-wide wait) when it has to grow; a stream
 * that is being captured gets no buffer (no grid: the chain kernel alone).  The `_ws` twins below take the grid's memory
 * from the caller (sph3d_build_sphere_neighbor_workspace bytes, 16-byte aligned): they never allocate, never synchronise and
 * hold no state between calls — what SURVEY 8b asks of every kernel entry; the Python ops of this package call only those.
 * workspace == NULL there: no grid.  sph3d_release_stream_scratch(stream) frees the convenience buffer of `stream` on the
 * current device (call it before destroying the stream), sph3d_release_all_scratch() every buffer of the current device;
 * both return the number of buffers freed.  Environment SPH3D_NNGRID=0 (read once) turns the grid off. */
#define SPH3D_MAX_GROWTH_PASSES 4096
/* diagnostic: calls so far in this process whose early positions went through the cell grid */
long long sph3d_nngrid_launches(void);
int sph3d_build_sphere_neighbor(int B, int N, int M, int nn_sample, float radius,
                                const float* database, const float* query,
                                int* nn_index, int* nn_count, float* nn_dist,
                                sph3d_stream_t stream);
/* Same search with a FIXED radius per query — NOT the reference's semantics: every query starts from `radius` (growth by
 * 0.05 only until it has one neighbour; nothing is carried along the reference-thread chain or from cloud to cloud).
 * For clouds much larger than the reference's 8192-point blocks (BASELINE config 5: 65 536 points), where the chain
 * lets the radius reach metres and saturates every row at K. */
int sph3d_build_sphere_neighbor_fixed(int B, int N, int M, int nn_sample, float radius,
                                const float* database, const float* query,
                                int* nn_index, int* nn_count, float* nn_dist,
                                sph3d_stream_t stream);

/* The same two searches with the cell grid's memory from the caller (see above; tf_nnquery.cpp:53-54: the op's Compute
 * allocates every buffer the launcher touches — this is that contract). */
size_t sph3d_build_sphere_neighbor_workspace(int B, int N, int M);
int sph3d_build_sphere_neighbor_ws(int B, int N, int M, int nn_sample, float radius,
                                   const float* database, const float* query,
                                   int* nn_index, int* nn_count, float* nn_dist,
                                   void* workspace, size_t workspace_bytes, sph3d_stream_t stream);
int sph3d_build_sphere_neighbor_fixed_ws(int B, int N, int M, int nn_sample, float radius,
                                   const float* database, const float* query,
                                   int* nn_index, int* nn_count, float* nn_dist,
                                   void* workspace, size_t workspace_bytes, sph3d_stream_t stream);
int sph3d_release_stream_scratch(sph3d_stream_t stream);
int sph3d_release_all_scratch(void);

/* replaces buildCubeNeighborLauncher (tf_nnquery_gpu.cu:123-127; kernel
 * cal_nn_binidx_cube :72-113; op BuildCubeNeighbor tf_nnquery.cpp:116-168).
 * -> nn_index[B,M,K,2] i32 = (database index, cubic bin id), nn_count[B,M]. */
int sph3d_build_cube_neighbor(int B, int N, int M, int grid_size, int nn_sample, float length,
                              const float* database, const float* query,
                              int* nn_index, int* nn_count,
                              sph3d_stream_t stream);

/* Fused graph construction of one level: neighbour search + spherical-kernel bins in ONE kernel, and — when
 * transpose_workspace is given (sph3d_graph_transpose_workspace bytes) — the counting pass of the transposed graph as
 * well (finish it with sph3d_graph_transpose_finish).  Outputs equal, bit for bit, sph3d_build_sphere_neighbor followed by
 * sph3d_spherical_kernel(n, p, q, radius) on the same database / query; shapes whose per-query hit lists do not fit LDS
 * run those kernels one after the other.  filt_index == NULL: no bins (an inter-level graph, n / p / q ignored): the
 * search plus the counting pass for F = 1 (transpose_workspace is then required). */
int sph3d_build_sphere_graph(int B, int N, int M, int nn_sample, float radius, int n, int p, int q,
                             const float* database, const float* query,
                             int* nn_index, int* nn_count, float* nn_dist, int* filt_index,
                             void* transpose_workspace, size_t transpose_workspace_bytes, sph3d_stream_t stream);
/* The same with ROCm's device-library atan2f for the bins (= sph3d_build_sphere_neighbor + sph3d_spherical_kernel_ocml:
 * bit for bit the bins of the reference's own kernel built for this GPU; not reproducible on a CPU). */
int sph3d_build_sphere_graph_ocml(int B, int N, int M, int nn_sample, float radius, int n, int p, int q,
                             const float* database, const float* query,
                             int* nn_index, int* nn_count, float* nn_dist, int* filt_index,
                             void* transpose_workspace, size_t transpose_workspace_bytes, sph3d_stream_t stream);

/* Both of the above with the search's cell grid from the caller (search_workspace: sph3d_build_sphere_neighbor_workspace(B, N, M)
 * bytes, NULL = no grid): no allocation, legal under stream capture.  ocml: 0 = sph3d_build_sphere_graph, 1 = ..._ocml. */
int sph3d_build_sphere_graph_ws(int B, int N, int M, int nn_sample, float radius, int n, int p, int q, int ocml,
                                const float* database, const float* query,
                                int* nn_index, int* nn_count, float* nn_dist, int* filt_index,
                                void* transpose_workspace, size_t transpose_workspace_bytes,
                                void* search_workspace, size_t search_workspace_bytes, sph3d_stream_t stream);

/* ---- buildkernel --------------------------------------------------------
 * replaces sphericalKernelLauncher (tf_ops/buildkernel/tf_buildkernel_gpu.cu:83-89;
 * kernel build_spherical_kernel :20-79; op SphericalKernel tf_buildkernel.cpp:35-99).
 * -> filt_index[B,M,K] i32: 0 = self / unused slot, else
 * 1 + nID + n*pID + n*p*qID.  Requires n>2 even, p>0 even, q>0 (:39-49).
 * atan2f is include/sph3d_atan2f.h (correctly rounded, the same bits on device and in the CPU oracle).
 * sph3d_spherical_kernel_ocml is the same kernel calling ROCm's device-library atan2f: bit-for-bit the bins the
 * reference's own kernel produces when built for this GPU; they differ from the default only for neighbours within an
 * ulp of an angular bin boundary and are not reproducible on a CPU. */
int sph3d_spherical_kernel(int B, int N, int M, int K, int n, int p, int q, float radius,
                           const float* database, const float* query,
                           const int* nn_index, const int* nn_count, const float* nn_dist,
                           int* filt_index,
                           sph3d_stream_t stream);
int sph3d_spherical_kernel_ocml(int B, int N, int M, int K, int n, int p, int q, float radius,
                                const float* database, const float* query,
                                const int* nn_index, const int* nn_count, const float* nn_dist,
                                int* filt_index,
                                sph3d_stream_t stream);

/* ---- convolution --------------------------------------------------------
 * replaces depthwiseConv3dLauncher (tf_ops/convolution/tf_conv3d_gpu.cu:107-113;
 * kernel depthwise_conv3d_forward :7-29; op DepthwiseConv3d tf_conv3d.cpp:47-94).
 * out[b,m,c*r+rho] = (1/cnt) * sum_k in[b,idx_k,c] * filt[bin_k,c,rho].
 * F (= filter.shape[0]) is an extra argument: the filter table is staged in LDS. */
int sph3d_depthwise_conv3d(int B, int N, int M, int F, int C, int r, int K,
                           const int* nn_index, const int* nn_count, const int* bin_index,
                           const float* input, const float* filter, float* output,
                           sph3d_stream_t stream);

/* replaces depthwiseConv3dGradLauncher (tf_conv3d_gpu.cu:115-140; kernels
 * depthwise_input_backward :32-55, depthwise_filter_backward :58-101;
 * op DepthwiseConv3dGrad tf_conv3d.cpp:101-158).
 * grad_input[B,N,C], grad_filter[F,C,r] are fully written.  The wrapper builds the transposed graph
 * (below) in `workspace` (sph3d_depthwise_conv3d_grad_workspace() bytes of device memory) and runs
 * sph3d_depthwise_conv3d_grad_t; callers that keep the transpose across calls use that directly. */
size_t sph3d_depthwise_conv3d_grad_workspace(int B, int N, int M, int F, int C, int r, int K);
int sph3d_depthwise_conv3d_grad(int B, int N, int M, int F, int C, int r, int K,
                                const int* nn_index, const int* nn_count, const int* bin_index,
                                const float* input, const float* filter, const float* grad_output,
                                float* grad_input, float* grad_filter,
                                void* workspace, size_t workspace_bytes,
                                sph3d_stream_t stream);

/* ---- transposed neighbour graph (not in the reference: how its atomic scatters are avoided) ----
 * Every gradient of the path is a scatter over the neighbour graph; libsph3d runs it as a gather over
 * the graph's transpose ("in-edge lists"), built once per graph and reusable by every gradient that
 * uses the same (nn_index, nn_count[, bin_index | weight]).  F = number of filter bins (1 when
 * bin_index is NULL).  In-edges are sorted by (cloud, source point n, bin f):
 *   offsets[B*(N*F+1)]  edges of segment (b,n,f) are entries [offsets[s], offsets[s+1]), s = b*(N*F+1) + n*F + f
 *   ent_key[B*M*K]      m, the graph row (output point) the edge comes from
 *   ent_scale[B*M*K]    1/nn_count[b,m]   (weight[b,m,k] when weight is not NULL)
 * workspace: sph3d_graph_transpose_workspace() bytes of scratch (segment counters). */
size_t sph3d_graph_transpose_workspace(int B, int N, int M, int K, int F);
int sph3d_graph_transpose(int B, int N, int M, int K, int F,
                          const int* nn_index, const int* nn_count,
                          const int* bin_index /* or NULL */, const float* weight /* or NULL */,
                          int* offsets, int* ent_key, float* ent_scale,
                          int* active_bins /* [F+1] or NULL: count, then the ascending list of the bins that occur */,
                          void* workspace, size_t workspace_bytes, sph3d_stream_t stream);
/* PACKED entries (round 6): ent_scale == NULL asks sph3d_graph_transpose[_finish[_ordered]] for entry words
 * ent_key[e] = m | nn_count[m] << 24 and no scale array — allowed for un-weighted graphs (weight == NULL) with M <= 2^24 and K <= 255
 * (SPH3D_EINVAL otherwise); every consumer below that takes (ent_key, ent_scale) accepts ent_scale == NULL for such a graph and
 * derives 1 / nn_count[m] from the word (the same correctly rounded division); sph3d_max_pool3d_grad_t masks the row bits itself.
 * The two phases of sph3d_graph_transpose on one workspace: segment counts (also produced by sph3d_build_sphere_graph), then
 * scan + fill. */
int sph3d_graph_transpose_count(int B, int N, int M, int K, int F, const int* nn_index, const int* nn_count,
                                const int* bin_index, int want_active, void* workspace, size_t workspace_bytes,
                                sph3d_stream_t stream);
int sph3d_graph_transpose_finish(int B, int N, int M, int K, int F,
                                 const int* nn_index, const int* nn_count, const int* bin_index,
                                 const float* weight, int* offsets, int* ent_key, float* ent_scale, int* active_bins,
                                 void* workspace, size_t workspace_bytes, sph3d_stream_t stream);
/* The pooling graph of a level = the rows of its intra-level graph at the sampled points (the two tf.gather_nd of
 * models/SPH3D_s3dis.py:68-72) in ONE launch that, with a transpose workspace (sph3d_graph_transpose_workspace(B, N, S, K, 1) bytes;
 * NULL: copy only), also runs the counting phase of the pooling graph's transposed graph (finish it with
 * sph3d_graph_transpose_finish[_ordered]).  pairs[B*S][2] = (cloud, point); out_index [B, S, K], out_count [B, S]. */
int sph3d_gather_rows_count(int B, int N, int S, int K, const int* pairs, const int* nn_index, const int* nn_count,
                            int* out_index, int* out_count, void* transpose_workspace, size_t transpose_workspace_bytes,
                            sph3d_stream_t stream);
/* sph3d_graph_transpose_finish that also writes sph3d_graph_balanced_order's permutation (order[B*N]; NULL: none) from inside its
 * fill launch: scan (one pass) + fill/order = two launches per graph. */
int sph3d_graph_transpose_finish_ordered(int B, int N, int M, int K, int F,
                                         const int* nn_index, const int* nn_count, const int* bin_index,
                                         const float* weight, int* offsets, int* ent_key, float* ent_scale, int* active_bins,
                                         int* order, void* workspace, size_t workspace_bytes, sph3d_stream_t stream);
/* A processing order for sph3d_depthwise_conv3d_grad_t (its source_order argument) that balances the in-edges over the
 * gradient kernel's waves: inside windows of 2048 consecutive source points the points are sorted by in-degree (read from
 * `offsets` of the transposed graph with F bins), alternately descending and ascending.  order[B*N], a permutation per
 * cloud; results of the gradient are the same sums in a different order of the filter-gradient partials. */
int sph3d_graph_balanced_order(int B, int N, int F, const int* offsets, int* order, sph3d_stream_t stream);
/* conv gradients from a prebuilt transposed graph: both gradients in one pass, no float atomics
 * (grad_input gathered in registers; grad_filter accumulated in per-lane registers by persistent workgroups
 * that sweep the clouds of their XCD, one partial table per workgroup written to `workspace` =
 * sph3d_depthwise_conv3d_grad_t_workspace() bytes, then reduced). */
size_t sph3d_depthwise_conv3d_grad_t_workspace(int B, int N, int F, int C, int r);
int sph3d_depthwise_conv3d_grad_t(int B, int N, int M, int F, int C, int r,
                                  const int* offsets, const int* ent_key, const float* ent_scale,
                                  const int* source_order /* optional [B,N] permutation per cloud: the order in which the sources are swept (NULL = index order) */,
                                  const int* active_bins /* optional, from sph3d_graph_transpose: lets a graph with <= 17 occurring bins run with half the accumulator registers */,
                                  const float* input, const float* filter, const float* grad_output,
                                  float* grad_input, float* grad_filter,
                                  void* workspace, size_t workspace_bytes, sph3d_stream_t stream);
/* (transposed graph built with F = 1)  grad_input[B,Nin,C] = sum over in-edges of grad_output[B,Mout,C] * ent_scale: the gradient of
 * avg_pool3d (Nin=N, Mout=M), mean_interpolate and weighted_interpolate (Nin=M coarse, Mout=N fine). */
int sph3d_scatter_grad_t(int B, int Nin, int Mout, int C,
                         const int* offsets, const int* ent_key, const float* ent_scale,
                         const float* grad_output, float* grad_input, sph3d_stream_t stream);
/* bytes of workspace the *_grad wrappers below need to build the transposed graph internally
 * (N = number of SOURCE points of the gradient, M = number of graph rows) */
size_t sph3d_scatter_grad_workspace(int B, int N, int M, int K);

/* ---- pooling ------------------------------------------------------------
 * replaces maxPool3dLauncher / maxPool3dGradLauncher / avgPool3dLauncher /
 * avgPool3dGradLauncher (tf_ops/pooling/tf_pool3d_gpu.cu:93-119; kernels :5-90;
 * ops tf_pool3d.cpp:63-234).  max_index[B,M,C] holds the argmax POINT id
 * (first maximum wins, strict >). */
int sph3d_max_pool3d(int B, int N, int M, int C, int K,
                     const int* nn_index, const int* nn_count, const float* input,
                     float* output, int* max_index, sph3d_stream_t stream);
int sph3d_max_pool3d_grad(int B, int N, int M, int C,
                          const int* max_index, const float* grad_output,
                          float* grad_input, sph3d_stream_t stream);
/* the same gradient as a gather over the TRANSPOSED pooling graph (offsets / ent_key of sph3d_graph_transpose with F = 1 on
 * the pooling graph's nn_index / nn_count): no memset, no float atomics, fixed summation order.  nn_count [B,M]: rows
 * without neighbours carry max_index 0 and their gradient goes to point 0, as in the scatter form. */
int sph3d_max_pool3d_grad_t(int B, int N, int M, int C, const int* offsets, const int* ent_key, const int* nn_count,
                            const int* max_index, const float* grad_output,
                            const float* addend /* optional [B,N,C]: another gradient of the same input, added in (NULL: none) */,
                            float* grad_input, sph3d_stream_t stream);
int sph3d_avg_pool3d(int B, int N, int M, int C, int K,
                     const int* nn_index, const int* nn_count, const float* input,
                     float* output, sph3d_stream_t stream);
int sph3d_avg_pool3d_grad(int B, int N, int M, int C, int K,
                          const int* nn_index, const int* nn_count, const float* grad_output,
                          float* grad_input,
                          void* workspace /* sph3d_scatter_grad_workspace(B,N,M,K) */, size_t workspace_bytes,
                          sph3d_stream_t stream);

/* ---- unpooling ----------------------------------------------------------
 * replaces meanInterpolateLauncher / meanInterpolateGradLauncher /
 * weightedInterpolateLauncher / weightedInterpolateGradLauncher
 * (tf_ops/unpooling/tf_unpool3d_gpu.cu:87-113; kernels :5-84; ops
 * tf_unpool3d.cpp:64-242).  Here N = fine/output count, M = coarse/input
 * count (the reference's naming): input[B,M,C], nn_index[B,N,K] -> output[B,N,C]. */
int sph3d_mean_interpolate(int B, int N, int M, int C, int K,
                           const int* nn_index, const int* nn_count, const float* input,
                           float* output, sph3d_stream_t stream);
int sph3d_mean_interpolate_grad(int B, int N, int M, int C, int K,
                                const int* nn_index, const int* nn_count, const float* grad_output,
                                float* grad_input,
                                void* workspace /* sph3d_scatter_grad_workspace(B,M,N,K) */, size_t workspace_bytes,
                                sph3d_stream_t stream);
int sph3d_weighted_interpolate(int B, int N, int M, int C, int K,
                               const int* nn_index, const int* nn_count,
                               const float* input, const float* weight,
                               float* output, sph3d_stream_t stream);
int sph3d_weighted_interpolate_grad(int B, int N, int M, int C, int K,
                                    const int* nn_index, const int* nn_count,
                                    const float* grad_output, const float* weight,
                                    float* grad_input,
                                    void* workspace /* sph3d_scatter_grad_workspace(B,M,N,K) */, size_t workspace_bytes,
                                    sph3d_stream_t stream);

/* ---- sampling -----------------------------------------------------------
 * replaces farthestPointSampleLauncher (tf_ops/sampling/tf_sample_gpu.cu:77-80;
 * kernel farthestpointsampleKernel :7-73; op FarthestPointSample tf_sample.cpp:30-58).
 * inp[b,n,3] -> out[b,m] i32; index 0 first, then argmax of the running
 * minimum squared distance; ties: lower (k mod 1024) wins, then lower k
 * (the reference's thread/tree order, :49,:56-66).  Race-free (the reference
 * has a latent race at :68).  Clouds of 2049 .. 16384 points run a PRUNED form of
 * the same chain (csrc/sample.hip: fps_prune_kernel): spatially sorted blobs of 64
 * points whose distance update is skipped when the new sample provably cannot
 * lower any of their running distances; same samples bit for bit.
 * workspace: sph3d_farthest_point_sample_workspace(b,n,m) bytes (0 when the
 * cloud fits the register-resident kernels, n <= 24576). */
size_t sph3d_farthest_point_sample_workspace(int b, int n, int m);
int sph3d_farthest_point_sample(int b, int n, int m, const float* inp, int* out,
                                void* workspace, size_t workspace_bytes,
                                sph3d_stream_t stream);

/* tf.gather_nd with [.., 2] (cloud, point) index pairs, as the model graphs use it on coordinates, neighbour lists and
 * counts of the sampled points (models/SPH3D_s3dis.py:68-72; TensorFlow's stock op in the reference): params is
 * [B, N, row] 4-byte elements, pairs [S, 2] int32, out [S, row].  Out-of-range pairs are clamped. */
int sph3d_gather_nd(int B, int N, long long S, int row, const int* pairs, const void* params, void* out,
                    sph3d_stream_t stream);

/* A permutation of each cloud in which consecutive points are close in space (counting sort by the Morton code of a point's
 * cell in a grid over the cloud's bounding box): order[B,N] int32.  A processing order for gather kernels — it never changes a
 * result, only which rows a workgroup touches together. */
int sph3d_spatial_order(int B, int N, const float* xyz, int* order, sph3d_stream_t stream);

/* ---- pointwise 1x1 feature GEMM (fp32 MFMA) -----------------------------
 * replaces the tf.matmul inside separable_conv3d / pointwise_conv3d /
 * fully_connected (utils/sph3gcn_util.py:146-150, 204-206, 260) -> cuBLAS SGEMM
 * in the reference.  Y[R,Cout] = X[R,Cin] * W[Cin,Cout] (+ bias[Cout]) with an
 * optional fused ELU; exact fp32 (v_mfma_f32_32x32x2_f32).
 * act: 0 = none, 1 = ELU.  bias may be NULL.
 * trans_x / trans_w select the backward products:
 *   dX[R,Cin]  = dY[R,Cout] * W^T      -> sph3d_pointwise_gemm(R, Cout, Cin, dY, W, trans_w=1)
 *   dW[Cin,Cout] = X^T[Cin,R] * dY     -> sph3d_pointwise_gemm_tn(...) */
int sph3d_pointwise_gemm(int R, int Cin, int Cout,
                         const float* X, const float* W, const float* bias, int act,
                         int trans_w, float* Y, sph3d_stream_t stream);
/* How the whole-tile products (all three, and the statistics variant) run — process-wide, returns the previous mode:
 *   1 (default): bf16 matrix pipe with every fp32 operand cut EXACTLY into three bf16 pieces and the six leading piece products
 *      accumulated in fp32 (csrc/gemm.hip: gemm_split_mfma) — fp32-sized error (<= 2^-23 per product), 2.7x the fp32 MFMA rate;
 *   0: v_mfma_f32_32x32x2_f32, a k-ordered fp32 fmaf chain (what ragged shapes always take);  other values: query only. */
int sph3d_pointwise_gemm_mode(int mode);
/* Products whose tile grid is too small to fill the chip run several workgroups per tile that exchange their partial accumulators inside
 * the launch (csrc/gemm.hip); the waiting workgroup's spin is bounded.  -> number of launches that ever gave up waiting
 * (never expected: 0; synchronises with the device; -1 if the counter cannot be read). */
int sph3d_pointwise_gemm_exchange_failures(void);
int sph3d_pointwise_gemm_tn(int R, int Cin, int Cout,
                            const float* X, const float* dY, float* dW,
                            void* workspace, size_t workspace_bytes,
                            sph3d_stream_t stream);
size_t sph3d_pointwise_gemm_tn_workspace(int R, int Cin, int Cout);

/* ---- fused ELU + batch-norm tail of separable_conv3d / pointwise_conv3d --------------------------
 * replaces tf.nn.elu + tf.layers.batch_normalization(momentum=0.99, epsilon=1e-3) applied after the
 * pointwise matmul (utils/sph3gcn_util.py:152-161, 208-220, 328-332; stock TF ops in the reference).
 * y[R,C] is the matmul output; out = (elu(y) - mean) * rstd * gamma + beta with per-channel statistics of
 * elu(y) over the R rows (training) or the running statistics (inference).  `momentum` is the weight of
 * the NEW batch statistics (1 - 0.99).  save_mean / save_rstd [C] feed the backward pass, which needs only
 * y (elu(y) is recomputed).  Requires C % 4 == 0 and C <= 1024.  workspace: sph3d_elu_bn_workspace(). */
size_t sph3d_elu_bn_workspace(int R, int C);
int sph3d_elu_bn_forward(int R, int C, const float* y, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, int training,
                         float* out, float* save_mean, float* save_rstd,
                         void* workspace, size_t workspace_bytes, sph3d_stream_t stream);
int sph3d_elu_bn_backward(int R, int C, const float* y, const float* dout, const float* gamma,
                          const float* save_mean, const float* save_rstd, int training,
                          float* dy, float* dgamma, float* dbeta,
                          void* workspace, size_t workspace_bytes, sph3d_stream_t stream);

/* ---- fused forward of a layer tail: GEMM whose epilogue emits the batch-norm statistics' partial sums ---------------
 * (utils/sph3gcn_util.py:146-161: tf.matmul -> ELU -> tf.layers.batch_normalization; SURVEY 8f.3, first half.)
 * sph3d_pointwise_gemm_bnstats writes Y = X W (+ bias; raw, the backward pass needs it) and partial[nblk][2][Cout] = per row block
 * sum elu(y), sum elu(y)^2; nblk = sph3d_pointwise_gemm_bnstats_blocks(R, Cin, Cout), 0 when the shape does not qualify
 * (whole 128/64-row tiles, Cout a multiple of 64 — of 128 above 64 —, Cin a multiple of 16; 16-byte aligned operands):
 * then run sph3d_pointwise_gemm and sph3d_elu_bn_forward.  sph3d_elu_bn_forward_partials is sph3d_elu_bn_forward
 * (training mode) without its statistics pass over Y. */
int sph3d_pointwise_gemm_bnstats_blocks(int R, int Cin, int Cout);
int sph3d_pointwise_gemm_bnstats(int R, int Cin, int Cout, const float* X, const float* W, const float* bias /* [Cout] or NULL */,
                                 float* Y, float* partial, sph3d_stream_t stream);
int sph3d_elu_bn_forward_partials(int R, int C, int nblk, const float* partial, const float* y, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                  float* out, float* save_mean, float* save_rstd, sph3d_stream_t stream);

/* ---- fused separable convolution for inference ---------------------------------------------------------
 * The whole layer of utils/sph3gcn_util.py:134-161 with is_training=False in one kernel: DepthwiseConv3d
 * (tf_conv3d.cpp:34-107 / tf_conv3d_gpu.cu:7-29) -> matmul with pointwise_weights[C*r][Cout] -> + bias[Cout] (NULL: none)
 * -> act (0 none | 1 ELU) -> y*scale[Cout] + shift[Cout] (NULL: identity; batch norm with the moving statistics is
 * scale = gamma / sqrt(moving_var + eps), shift = beta - moving_mean*scale).  output [B, M, Cout]; the [B, M, C*r] depthwise
 * tensor is never written.  Shapes: r in {1, 2}, C <= 128 a multiple of 4, Cout <= 128 a multiple of 16, F <= 254;
 * sph3d_separable_conv3d_fused_supported() -> 1 | 0, and the call returns SPH3D_EUNSUPPORTED outside them (run
 * sph3d_depthwise_conv3d_forward + sph3d_pointwise_gemm then). */
int sph3d_separable_conv3d_fused_supported(int N, int F, int C, int r, int K, int Cout);
int sph3d_separable_conv3d_fused(int B, int N, int M, int F, int C, int r, int K, int Cout, int act,
                                 const int* nn_index, const int* nn_count, const int* bin_index, const float* input,
                                 const float* depthwise_filter, const float* pointwise_weights, const float* bias,
                                 const float* scale, const float* shift, float* output, sph3d_stream_t stream);

/* ---- fused separable convolution for TRAINING ("BN stats as a side reduction") ----------------------------------
 * The layer of utils/sph3gcn_util.py:134-161 with is_training=True up to the batch-norm statistics, in one barrier-free kernel
 * (csrc/sepring.hip): DepthwiseConv3d (tf_conv3d.cpp:34-107 / tf_conv3d_gpu.cu:7-29) -> matmul with
 * pointwise_weights[C*r][Cout] (+ bias[Cout], NULL: none).  Writes
 *   depthwise_output [B, M, C*r]   the depthwise tensor (operand of the weight gradient; not re-read by the forward pass),
 *   y                [B, M, Cout]  the raw product (what sph3d_elu_bn_backward needs),
 *   partial          [nblk][2][Cout]  per (workgroup, wave group) sum elu(y), sum elu(y)^2 over its rows — the layout
 *                    sph3d_elu_bn_forward_partials consumes; nblk = sph3d_separable_conv3d_train_blocks(Cout).
 * Shapes: r in {1, 2}, C % 4 == 0, C <= 128, C*r <= 256, Cout in {16, 32, 64, 128, 256}, F <= 254
 * (sph3d_separable_conv3d_train_supported() -> 1 | 0; SPH3D_EUNSUPPORTED otherwise: run sph3d_depthwise_conv3d +
 * sph3d_pointwise_gemm_bnstats).  The same kernel without the two extra outputs serves sph3d_separable_conv3d_fused on
 * these shapes.  sph3d_separable_conv3d_ring_failures(): 1 if a launch ever gave up waiting (never expected; synchronises). */
int sph3d_separable_conv3d_train_supported(int N, int F, int C, int r, int K, int Cout);
int sph3d_separable_conv3d_train_blocks(int Cout);
int sph3d_separable_conv3d_train(int B, int N, int M, int F, int C, int r, int K, int Cout,
                                 const int* nn_index, const int* nn_count, const int* bin_index, const float* input,
                                 const float* depthwise_filter, const float* pointwise_weights, const float* bias,
                                 float* depthwise_output, float* y, float* partial, sph3d_stream_t stream);
int sph3d_separable_conv3d_ring_failures(void);

/* ---- the 1x1 layer with few outputs over two operand halves (the logits layer) ------------------------------------------------
 * tf.concat((unpooled, skip), axis=2) followed by pointwise_conv3d to num_cls outputs (models/SPH3D_s3dis.py:104-108,
 * utils/sph3gcn_util.py:166-222) without materialising the concatenation:
 *   sph3d_pointwise_gemm_skinny     Y[R,N] = A1[R,K1] W[0:K1] + A2[R,K2] W[K1:K1+K2] (+ bias[N] or NULL); A2 = NULL with K2 = 0: one operand
 *   sph3d_pointwise_gemm_skinny_tn  dW[K1+K2,N] = [A1 | A2]^T dY   (workspace: ..._tn_workspace bytes; fixed summation order)
 * N <= 16, K1 and K2 multiples of 16, K1 + K2 <= 256 in at most four started groups of 64 channels (ceil(K1/64) + ceil(K2/64) <= 4),
 * 16-byte aligned operands (..._supported() -> 1 | 0; SPH3D_EUNSUPPORTED
 * otherwise: concatenate and call sph3d_pointwise_gemm / _tn).  The input gradients are sph3d_pointwise_gemm(trans_w = 1) with
 * the matching rows of W, once per operand half. */
int sph3d_pointwise_gemm_skinny_supported(int R, int K1, int K2, int N);
int sph3d_pointwise_gemm_skinny(int R, int K1, int K2, int N, const float* A1, const float* A2, const float* W, const float* bias,
                                float* Y, sph3d_stream_t stream);
size_t sph3d_pointwise_gemm_skinny_tn_workspace(int R, int K1, int K2, int N);
int sph3d_pointwise_gemm_skinny_tn(int R, int K1, int K2, int N, const float* A1, const float* A2, const float* dY, float* dW,
                                   void* workspace, size_t workspace_bytes, sph3d_stream_t stream);

/* ---- the depthwise convolution over a channel concatenation that is never materialised --------------------------------------
 * DepthwiseConv3d / DepthwiseConv3dGrad (tf_conv3d.cpp:34-107, :109-205) applied to tf.concat((input_a, input_b), axis=2)
 * (models/SPH3D_s3dis.py:100-104: a decoder level's un-pooled features and the encoder's skip features): input_a [B,N,Ca],
 * input_b [B,N,Cb], filter [F, Ca+Cb, r], output [B,M,(Ca+Cb)*r]; the gradient writes grad_a [B,N,Ca] and grad_b [B,N,Cb]
 * (workspace: sph3d_depthwise_conv3d_grad_t_workspace(B, N, F, Ca+Cb, r)).  The kernels' channel slices are 256 outputs wide and
 * must lie inside one input: Ca*r a multiple of 256, Ca+Cb > 128, (Ca+Cb) % 4 == 0, r in {1,2}, F <= 33
 * (..._cat_supported() -> 1 | 0; SPH3D_EUNSUPPORTED otherwise: concatenate and call the plain entries). */
int sph3d_depthwise_conv3d_cat_supported(int F, int Ca, int Cb, int r);
int sph3d_depthwise_conv3d_cat(int B, int N, int M, int F, int Ca, int Cb, int r, int K, const int* nn_index, const int* nn_count,
                               const int* bin_index, const float* input_a, const float* input_b, const float* filter, float* output,
                               sph3d_stream_t stream);
int sph3d_depthwise_conv3d_grad_t_cat(int B, int N, int M, int F, int Ca, int Cb, int r, const int* offsets, const int* ent_key,
                                      const float* ent_scale, const int* source_order, const int* active_bins, const float* input_a,
                                      const float* input_b, const float* filter, const float* grad_output, float* grad_a,
                                      float* grad_b, float* grad_filter, void* workspace, size_t workspace_bytes,
                                      sph3d_stream_t stream);

/* ---- the training loop's optimiser update (train_s3dis.py:224, tf.train.AdamOptimizer) over flat fp32 buffers: one streaming
 * pass, torch.optim.Adam's arithmetic (no amsgrad / weight decay); step = 1, 2, ... (bias corrections are computed on the host) */
int sph3d_adam_step(long long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                    float beta2, float eps, int step, sph3d_stream_t stream);

/* the segmentation nets' training loss (models/SPH3D_s3dis.py:116-133: per block the mean over the points with inner_label > 0
 * of the sparse softmax cross-entropy, summed over the batch by the caller) and its gradient in one launch:
 *   loss_part[b * S + s], S = sph3d_masked_softmax_xent_parts(N): the shares of S slices of block b's points in
 *       loss_b = mean_{n: inner > 0} ( logsumexp(logits[b,n,:]) - logits[b,n,label[b,n]] )      (0 for a block without such points)
 *   dlogits[b,n,:] = d(sum_b loss_b) / d logits[b,n,:]
 * logits [B,N,C] fp32, label [B,N] int64, inner_label [B,N] fp32; deterministic (fixed-order reductions).
 * dlogits == NULL: the losses only (evaluation: no gradient pass, no [B,N,C] store).  An inner point whose label lies
 * outside [0, C) makes its block's loss and its gradient row NaN (the reference's GPU op yields NaN there as well). */
int sph3d_masked_softmax_xent_parts(int N);
int sph3d_masked_softmax_xent(int B, int N, int C, const float* logits, const long long* label, const float* inner_label,
                              float* loss_part, float* dlogits, sph3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPH3D_H */
