/* lghip.h - C ABI of liblghip.so, the MI355X (gfx950) device library behind
 * lightgrad's HipTensor backend.
 *
 * This is the drop-in boundary: plain C, opaque device
 * pointers, caller-owned int64 shape/stride arrays that only need to live for
 * the duration of a call.  Every entry point is what a lightgrad backend binds
 * instead of the pyopencl calls of the reference's OpenCL backend; the
 * reference interface each one replaces is cited as file:line (relative to the
 * lightgrad repository).
 *
 * Conventions
 *   - every function returns 0 on success, a negative LG_E* code otherwise;
 *     lg_last_error() returns a thread-local message for the last failure.
 *     Nothing throws across the boundary.
 *   - one process drives one GPU through one HIP stream owned by the library
 *     (the reference: one in-order queue per device, opencl/device.py:79-84).
 *     All device work is stream-ordered and asynchronous; only lg_memcpy_d2h,
 *     lg_sync and lg_event_elapsed_ms block the host.  (The reference blocks
 *     after every kernel: opencl/kernels.py:194, :334, :499.)
 *   - shapes and strides are in ELEMENTS (not bytes), row-major order, at
 *     most LG_MAX_DIMS dimensions; a stride of 0 broadcasts that dimension.
 *   - arithmetic entry points are fp32 only; layout entry points (copy, fill)
 *     take an item size in bytes and are bit-exact for any dtype.
 */
#ifndef LGHIP_H
#define LGHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LG_MAX_DIMS 8

/* error codes */
#define LG_OK            0
#define LG_EINVAL       -1   /* bad argument (shape/stride/op id/alignment) */
#define LG_EHIP         -2   /* a HIP runtime call failed (message has hipGetErrorString) */
#define LG_ENOMEM       -3   /* device allocation failed even after trimming the pool */
#define LG_ENOTINIT     -4   /* lg_init has not been called */
#define LG_ECOMM        -5   /* RCCL failure (liblghip_comm.so) or a peer-window exchange that gave up waiting (lghip_p2p.h) */
#define LG_EINDEX       -6   /* an EARLIER kernel met an index / label out of range (reported by the next synchronising call) */

const char* lg_last_error(void);

/* ---- runtime: device, stream, pool ---------------------------------------
 * replaces OpenCLDevice (opencl/device.py:68-115): context + in-order queue +
 * cl.tools.MemoryPool(ImmediateAllocator) (:83-84). */
int lg_device_count(int* count);
int lg_init(int device);                       /* idempotent for the same device */
int lg_device(int* device);                    /* device bound by lg_init */
typedef struct {
    char     name[128];
    char     arch[32];                          /* "gfx950" */
    int32_t  compute_units;
    int32_t  clock_mhz;                         /* max engine clock */
    int32_t  wavefront_size;
    int32_t  lds_bytes_per_cu;
    uint64_t hbm_bytes;
    int32_t  l2_bytes;
    int32_t  reserved;
} lg_device_info_t;
int lg_device_info(lg_device_info_t* out);
/* Preflight of a multi-GPU job (no counterpart in the reference: its device pool enumerates OpenCL devices and never
 * relates two of them, opencl/device.py:12-49).  Pure queries, usable before lg_init and without touching `peer`:
 * can_access = hipDeviceCanAccessPeer(device, peer); link_type = hipExtGetLinkTypeAndHopCount's type (HSA_AMD_LINK_INFO_TYPE:
 * 0 HyperTransport, 1 QPI, 2 PCIe, 3 InfiniBand, 4 xGMI), hops its hop count; both -1 when the runtime cannot say. */
int lg_peer_info(int device, int peer, int* can_access, int* link_type, int* hops);
void* lg_stream(void);                          /* hipStream_t of the library (for liblghip_comm / profilers) */
int lg_sync(void);                              /* block until the stream is idle */

/* caching, stream-ordered allocator (replaces device.mem_pool.allocate, opencl/tensor.py:64).
 * lg_free returns the block to the pool immediately: safe because all users are on one stream. */
int lg_malloc(void** ptr, size_t bytes);
int lg_free(void* ptr);
int lg_pool_trim(void);                         /* hipFree every cached free block */
int lg_pool_stats(uint64_t* reserved_bytes, uint64_t* in_use_bytes, uint64_t* hip_malloc_calls);

/* transfers (replace cl.enqueue_copy: H2D opencl/tensor.py:78, D2H :84, D2D :92) */
int lg_memcpy_h2d(void* dst, const void* src, size_t bytes);   /* returns after src may be reused */
int lg_memcpy_d2h(void* dst, const void* src, size_t bytes);   /* synchronises the stream */
int lg_memcpy_d2d(void* dst, const void* src, size_t bytes);   /* stream-ordered */
/* upload without waiting: src is staged into pinned memory (reusable on return), the DMA is stream-ordered.
 * Feeds a new batch into the static input tensor of a captured graph between two lg_graph_launch calls. */
int lg_memcpy_h2d_async(void* dst, const void* src, size_t bytes);
/* HIP events on the library's stream (timing in bench.py) */
int lg_event_create(void** ev);
int lg_event_record(void* ev);
int lg_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */
int lg_event_destroy(void* ev);

/* hipGraph capture of the library's stream: a launch-bound step (the ~100
 * dispatches of one MLP training step) is recorded once and replayed with one
 * host call.  While capturing, pool blocks are pinned to the graph so replays
 * see the same addresses. */
int lg_graph_begin(void);
int lg_graph_end(void** graph_exec);
int lg_graph_launch(void* graph_exec);
int lg_graph_destroy(void* graph_exec);
int lg_graph_kernel_count(void* graph, int* kernels);          /* kernel launches one replay performs (reports, tests) */

/* ---- layout ops (bit-exact, any dtype) ----------------------------------- */

/* dst[idx] = src[idx] over `shape`; either side may be strided/broadcast.
 * replaces the `atom` kernel with op 'o = a' used by contiguous()/copy()/getitem/setitem
 * (opencl/tensor.py:103-116, opencl/ops.py:322-340). itemsize in {1,2,4,8}. */
int lg_copy_strided(int itemsize, int ndim, const int64_t* shape,
                    void* dst, const int64_t* dst_strides,
                    const void* src, const int64_t* src_strides);

/* dst[idx] = value over a strided view; `value_bits` holds the item in its low bytes.
 * replaces clEnqueueFillBuffer (opencl/ops.py:172-177). */
int lg_fill_strided(int itemsize, int ndim, const int64_t* shape,
                    void* dst, const int64_t* dst_strides, uint64_t value_bits);

/* ---- elementwise fp32 ------------------------------------------------------
 * One generic entry point replacing the run-time generated `atom` kernels
 * (opencl/kernels.py:24-195) for every op string used in opencl/ops.py:40-400.
 * Operands a, b, c, d are fp32 inputs with their own strides (0 = broadcast);
 * a NULL operand pointer means "this operand is the scalar `scalar`"
 * (reference: scalar kernel arguments, kernels.py:139-150).
 * out0/out1 are written; an output may alias an input with identical strides
 * (in-place forms `a += b`, kernels.py `additional_read`).
 * The library picks a vectorised contiguous path, an inner-dimension
 * vectorised path or a generic strided path; results do not depend on it. */
typedef enum {
    /* unary: out0 = f(a) */
    LG_EW_COPY = 0, LG_EW_NEG, LG_EW_EXP, LG_EW_LOG, LG_EW_RELU, LG_EW_SIGMOID,
    LG_EW_TANH, LG_EW_SIN, LG_EW_COS, LG_EW_SQRT,
    LG_EW_GELU,          /* tanh-approximated gelu of examples/bert.py:12, same evaluation order */
    /* binary: out0 = f(a, b) */
    LG_EW_ADD = 32, LG_EW_SUB, LG_EW_MUL, LG_EW_DIV, LG_EW_POW,
    LG_EW_RELU_BWD,      /* a = saved input t, b = g :  g * (t >= 0)        (cpu/ops.py:229) */
    LG_EW_SIGMOID_BWD,   /* a = y, b = g            :  y * (1 - y) * g      (cpu/ops.py:208) */
    LG_EW_TANH_BWD,      /* a = y, b = g            :  (1 - y*y) * g        (cpu/ops.py:219) */
    LG_EW_LOG_BWD,       /* a = x, b = g            :  (1 / x) * g          (cpu/ops.py:197) */
    LG_EW_SIN_BWD,       /* a = t, b = g            :  cos(t) * g           (cpu/ops.py:166) */
    LG_EW_COS_BWD,       /* a = t, b = g            :  -sin(t) * g          (cpu/ops.py:176) */
    LG_EW_EQ,            /* (a == b) ? 1 : 0 */
    LG_EW_GE,            /* (a >= b) ? 1 : 0 */
    LG_EW_BIAS_RELU,     /* a = x, b = bias         :  max(x + b, 0)  (fused Linear bias + relu) */
    LG_EW_GELU_BWD,      /* a = x, b = g            :  g * gelu'(x) */
    /* ternary: out0 = f(a, b, c) */
    LG_EW_MAX_BWD = 64,  /* a = x, b = extremum, c = g : g * (x == b)      (cpu/ops.py:272) */
    LG_EW_FMA,           /* a * b + c (two roundings, like the tape's mul then add) */
    /* two outputs: (out0, out1) = f(a, b, c[, d]) */
    LG_EW_MUL_BWD = 96,  /* a, b, c = g : (g * b, a * g)                   (cpu/ops.py:84)  */
    LG_EW_DIV_BWD,       /* a, b, c = g : (g / b, -a / b^2 * g)            (cpu/ops.py:94)  */
    LG_EW_POW_BWD        /* a, b, c = g, d = y : (b * a^(b-1) * g, g * y * log a) (cpu/ops.py:105) */
} lg_ew_op_t;

int lg_ew(int op, int ndim, const int64_t* shape,
          void* out0, const int64_t* out0_strides,
          void* out1, const int64_t* out1_strides,
          const void* a, const int64_t* a_strides,
          const void* b, const int64_t* b_strides,
          const void* c, const int64_t* c_strides,
          const void* d, const int64_t* d_strides,
          float scalar);

/* ---- reductions fp32 -------------------------------------------------------
 * replaces the multi-pass `reduce` kernel (opencl/kernels.py:344-501) as used by
 * sum/max/min (opencl/ops.py:344-400).  Dimensions whose bit is set in
 * `axis_mask` (bit i = dimension i) are reduced; `out` is CONTIGUOUS over the
 * kept dimensions in their original order.  max/min are exact; sum is a
 * tree reduction in fp32 (numpy uses pairwise blocks: agreement ~1e-6 rel). */
typedef enum { LG_RED_SUM = 0, LG_RED_MAX = 1, LG_RED_MIN = 2 } lg_red_op_t;
int lg_reduce(int op, int ndim, const int64_t* shape,
              const void* in, const int64_t* in_strides,
              uint32_t axis_mask, void* out);

/* ---- tensors that are not float32 (round 4) -----------------------------------
 * The reference's tensors keep whatever dtype their numpy array has (cpu/tensor.py:45-46: MNIST labels int16, data.py:43; BERT
 * ids int32, examples/bert.py:346) and its device backend generates every elementwise / reduction kernel per dtype
 * (opencl/kernels.py:9, :24-107, :344-431).  Here float32 - the north-star path - has its own tuned kernels (lg_ew, lg_reduce);
 * these entry points make int16 / int32 / int64 / float64 tensors computable on the device with numpy's semantics
 * (cpu/ops.py:52-84, :260-293): integers wrap around, integer sums are int64, max / min keep the dtype and propagate NaN.
 *   lg_ew_typed      out = a (op) b, all three of `dtype`, numpy broadcasting through strides; op = LG_EW_COPY / NEG (unary),
 *                    ADD / SUB / MUL, and for float64 DIV / POW; a NULL operand is the scalar (scalar_i for integers, scalar_f
 *                    for float64); out may alias an operand element for element (in-place forms)
 *   lg_reduce_typed  as lg_reduce; `out` holds int64 for integer sums, `dtype` otherwise
 *   lg_cast          out[i] = (dst type) in[i], any pair of the five dtypes (numpy's astype: float -> int truncates) */
typedef enum { LG_DT_I16 = 1, LG_DT_I32 = 2, LG_DT_I64 = 3, LG_DT_F64 = 4, LG_DT_F32 = 5 } lg_dtype_t;
int lg_ew_typed(int op, int dtype, int ndim, const int64_t* shape, void* out, const int64_t* out_strides,
                const void* a, const int64_t* a_strides, const void* b, const int64_t* b_strides,
                double scalar_f, int64_t scalar_i);
int lg_reduce_typed(int op, int dtype, int ndim, const int64_t* shape, const void* in, const int64_t* in_strides,
                    uint32_t axis_mask, void* out);
int lg_cast(int src_dtype, int dst_dtype, int ndim, const int64_t* shape, void* out, const int64_t* out_strides,
            const void* in, const int64_t* in_strides);

/* out (+)= reduction: with accumulate != 0 (sums only) the result is ADDED to `out`, so a bias gradient
 * can be accumulated straight into its gradient buffer (tensor.py:118 `grad += g` without the extra pass). */
int lg_reduce_acc(int op, int ndim, const int64_t* shape,
                  const void* in, const int64_t* in_strides,
                  uint32_t axis_mask, void* out, int accumulate);

/* ---- SGEMM on MFMA ---------------------------------------------------------
 * C[b] (M x N, row-major, leading dimension ldc) (+)= op(A[b]) @ op(B[b]), fp32
 * in/out, fp32 accumulate on v_mfma_f32_32x32x2_f32.
 *   transA = 0: A[m*lda + k]      transA = 1: A[k*lda + m]
 *   transB = 0: B[k*ldb + n]      transB = 1: B[n*ldb + k]
 * so stride-permuted views (W.T(1,0) in nn.Linear, nn.py:96; the transposed
 * operands of dot.backward, cpu/ops.py:116) are consumed WITHOUT the
 * contiguous() copies and zero-padding copies of the reference
 * (opencl/kernels.py:291-298, :319-320, :331).
 * batch > 1: operand b starts at base + b*stride{A,B,C} elements (stride 0
 * broadcasts an operand over the batch).  accumulate != 0 adds into C.
 * replaces kernels.dot (opencl/kernels.py:201-337) called from opencl/ops.py:116-132.  * Operands are fetched in 16-byte pieces along their contiguous index: when that extent is not a multiple of 4, up to
 * 12 bytes behind an operand's last row are READ (never used).  Memory from lg_malloc always has them; foreign
 * memory passed here must be readable that far.
 */
int lg_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                const float* A, int64_t lda, int64_t strideA,
                const float* B, int64_t ldb, int64_t strideB,
                float* C, int64_t ldc, int64_t strideC,
                int64_t batch, int accumulate);

/* Same product with a bias row added in the epilogue: C[b][m][n] = (op(A[b]) @ op(B[b]))[m][n] + bias[n]
 * (bias may be NULL).  The sum is rounded to fp32 before the bias is added, exactly like the separate
 * `x @ W.T + b` of nn.Linear (nn.py:96).  Fuses one elementwise pass per Linear layer. */
int lg_gemm_bias_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                     const float* A, int64_t lda, int64_t strideA,
                     const float* B, int64_t ldb, int64_t strideB,
                     float* C, int64_t ldc, int64_t strideC,
                     int64_t batch, const float* bias);

/* One matrix product with every epilogue / prologue the tape can fold into it (any of them may be off):
 *   bias [N]            added to every row (nn.py:96)                                      - excludes accumulate, rowsum
 *   rowsum [M]          (+)= row sums of op(A), see lg_gemm_rowsum_f32
 *   relu_a / relu_b     op(A) / op(B) pass through np.maximum(., 0) (cpu/ops.py:226) on their way to LDS: the consumer of
 *                       a relu reads the PRE-activation, the relu kernel and its output tensor are never made
 * (The (t >= 0) factor of relu's backward is NOT folded into the input-gradient GEMM: the relu output's own `.grad` -
 * readable on every tensor of the tape, tensor.py:40-42 - would then hold the masked values.) */
int lg_gemm_fused_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                      const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                      int accumulate, const float* bias, float* rowsum, int rowsum_accumulate,
                      int relu_a, int relu_b);

/* C = (op(A) @ op(B) [+ bias]) + addend, addend an [M, N] matrix with row pitch ldadd (>= N) that may be C itself only
 * through lg_gemm_f32's accumulate flag.  One matrix product.  The sums are formed in that order, each rounded to fp32 -
 * the values of the separate bias add and residual add (`dense(h) + h_in`, reference examples/bert.py:101; a gradient that
 * already holds a first contribution, autograd/tensor.py:111-118) without a pass over the result for either. */
int lg_gemm_addend_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                       const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                       const float* bias, const float* addend, int64_t ldadd);

/* One matrix product with an activation applied where the result is made (small tiles):
 *   act = LG_ACT_GELU      C = op(A) @ op(B) + bias (the pre-activation, kept for the backward) and aux[m][n] = gelu(C[m][n])
 *                          - nn.Linear followed by the gelu of examples/bert.py:12 without the elementwise pass
 *   act = LG_ACT_GELU_BWD  C = (op(A) @ op(B)) * gelu'(aux[m][n]), aux the pre-activation saved by the forward: the input
 *                          gradient of the NEXT Linear (dot.backward, cpu/ops.py:116) multiplied by the gelu derivative
 *                          in the same launch (bias must be NULL)
 * aux is [M, N] with row pitch ldaux.  The same expressions as the elementwise gelu kernels, bit for bit. */
#define LG_ACT_GELU 1
#define LG_ACT_GELU_BWD 2
int lg_gemm_act_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                    const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                    const float* bias, int act, float* aux, int64_t ldaux);

/* Three Linear layers on ONE input in one launch: C[i] = op(A) @ op(B[i]) + bias[i], i < 3, all of one shape - the query / key /
 * value projections of a transformer layer (reference examples/bert.py:78-80: three nn.Linear calls, each a kernels.dot and a
 * bias add).  The three weights stay where they are (separately allocated parameters; their addresses travel as offsets from
 * B[0]); the three results may be column blocks of one buffer (ldc = its row pitch).  bias may be NULL (no bias at all). */
int lg_gemm_multi3_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                       const float* const* B, int64_t ldb, float* const* C, int64_t ldc, const float* const* bias);

/* One product whose K dimension runs through three separately allocated right-hand operands of seg_k k-values each:
 * C (+)= op(A)[M, 3 * seg_k] @ [op(B[0]); op(B[1]); op(B[2])] (+ addend) - the input gradient of the three projections above,
 * dx = [dq | dk | dv] @ [Wq; Wk; Wv], which the tape forms as three kernels.dot calls and two additions (dot.backward,
 * cpu/ops.py:116; add_grad, tensor.py:111-118).  seg_k a multiple of 64; the sum runs in K order like one long product. */
int lg_gemm_kseg3_f32(int transA, int transB, int64_t M, int64_t N, int64_t seg_k, const float* A, int64_t lda,
                      const float* const* B, int64_t ldb, float* C, int64_t ldc, int accumulate,
                      const float* addend, int64_t ldadd);

/* lg_gemm_f32 over a TWO-level batch: matrix (o, i) of operand X starts at X + o*strideX_outer + i*strideX_inner.
 * One launch for attention-shaped products whose (batch, head) dims do not collapse into one stride after the head
 * split `reshape(b, s, h, d).transpose(0, 2, 1, 3)` (examples/bert.py:70-95; the reference's kernel is launched per
 * batch by kernels.dot, opencl/kernels.py:318-334). */
int lg_gemm_batched2_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                         const float* A, int64_t lda, int64_t strideA_outer, int64_t strideA_inner,
                         const float* B, int64_t ldb, int64_t strideB_outer, int64_t strideB_inner,
                         float* C, int64_t ldc, int64_t strideC_outer, int64_t strideC_inner,
                         int64_t batch_outer, int64_t batch_inner, int accumulate);

/* C (+)= op(A) @ op(B) and, from the same launch, rowsum (+)= row sums of op(A) (one matrix product, no batch).
 * With op(A) = g^T this is the weight gradient dW = g^T @ x together with the bias gradient db = column sums of g -
 * the dot.backward GEMM (cpu/ops.py:116) plus the un-broadcasting `sum(axis=0, keepdims=True)` of func.py:50-56 for
 * nn.Linear's `+ b` (nn.py:96).  The sums are column N of the product against a virtual column of ones appended to
 * op(B), i.e. they come off the same MFMA accumulators. */
int lg_gemm_rowsum_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                       const float* A, int64_t lda, const float* B, int64_t ldb,
                       float* C, int64_t ldc, int accumulate, float* rowsum, int rowsum_accumulate);

/* ---- fused optimizer ---------------------------------
 * One Adam/AdaBelief update of a contiguous parameter, numerically the
 * expression sequence of optim.py:36-40 / :48-52 evaluated per element (one
 * rounding per operation; the tape's `/` is multiply-by-reciprocal):
 *   g' = g * gscale (skipped when gscale == 1)
 *   m  = b1*m + (1-b1)*g';   s = belief ? g' - m : g';   v = b2*v + (1-b2)*s*s
 *   p += ((-lr) * (m * inv_bias1)) * (1 / (sqrt(v * inv_bias2) + eps))
 * with inv_bias1 = 1/(1 - b1^t), inv_bias2 = 1/(1 - b2^t) computed by the caller
 * in double; every scalar is rounded once to fp32 inside (numpy's python-float rule).
 * gscale folds the data-parallel 1/world_size into the update. */
int lg_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n,
                     double lr, double b1, double b2, double eps,
                     double inv_bias1, double inv_bias2, double gscale, int belief);

/* Graph-safe form: the optimizer step number is read from device memory (t = *step * t_mul + t_add,
 * the reference's per-PARAMETER `t`, optim.py:36/:48), so a captured hipGraph stays correct when
 * replayed; lg_counter_add_i64 advances the counter inside the same graph. */
int lg_adam_step_dev_f32(float* p, const float* g, float* m, float* v, int64_t n,
                         double lr, double b1, double b2, double eps,
                         const int64_t* step, int64_t t_mul, int64_t t_add, double gscale, int belief);
int lg_counter_add_i64(int64_t* counter, int64_t delta);

/* All parameters of a model in one launch (groups of 64 parameters per launch beyond that): p, g, m, v are flat buckets holding `nseg` parameters
 * back to back, parameter j occupying [offsets[j], offsets[j+1]); its step number is
 * t = (steps done) * nseg + j + 1.  Same arithmetic as lg_adam_step_dev_f32.
 *   step_slots == 0: step[0] holds the steps done and is only read (advance it with lg_counter_add_i64).
 *   step_slots  > 0: the launch advances the step number itself, with no hand-off between workgroups: `step` points at
 *     2 + step_slots int64 that all hold the steps done (zeros for a new optimizer); step[2 + s] is the private copy of
 *     workgroup s of the launch grid, step[0] the copy readers see (step[1] is unused).  step_slots >= nseg *
 *     ceil(longest parameter / 1024) always suffices. */
int lg_adam_multi_dev_f32(float* p, const float* g, float* m, float* v, int nseg, const int64_t* offsets,
                          double lr, double b1, double b2, double eps,
                          int64_t* step, int64_t step_slots, double gscale, int belief);

/* The update applied by the kernel that MAKES the gradient (round 4): `p += compute_delta(p.grad, i)` of optim.py:10-13 / :47-52
 * without a launch of its own.  The reference's optimizer is ~14 tensor expressions per parameter after backward; the fused
 * kernels above made that one launch; these entry points let the backward kernels themselves apply it to the values they are
 * about to store - GEMM epilogues (lg_gemm_f32 / lg_gemm_rowsum_f32 / lg_gemm_fused_f32 on the small tiles: C and the row
 * sums) and lg_head_bwd_f32 (dW, db).  Same arithmetic per element as lg_adam_multi_dev_f32, so the same bits.
 *   lg_adam_plan_create     one parameter's update, p_in -> p_out (p_out != p_in: other workgroups of the producing launch
 *                           still read the old values, e.g. dx = g @ W next to dW = g^T @ x), moments m, v in place, the
 *                           step number read from step_in[0] (t = step_in[0] * t_mul + t_add) and, for ONE plan of an
 *                           optimizer, written as step_out[0] = step_in[0] + 1 (step_out NULL elsewhere; step_out !=
 *                           step_in: an optimizer keeps two plans per parameter, A -> B and B -> A, and alternates).
 *   lg_adam_epilogue_arm    before backward: the next kernel that OVERWRITES `grad` (n floats, dense; beta = 0) applies the
 *                           plan; the gradient itself is still written.  A kernel that ADDS into `grad` after that is
 *                           refused (LG_EINVAL): for gradients made by ONE kernel per step.
 *   lg_adam_epilogue_finish after backward: applies the armed plans no kernel took, in one launch (none when all were
 *                           taken), and disarms everything.  Host bookkeeping only: a step recorded in a hipGraph replays
 *                           as recorded; the directions alternate, so a graph holds an EVEN number of steps. */
int lg_adam_plan_create(void** plan, const float* p_in, float* p_out, float* m, float* v, int64_t n,
                        const int64_t* step_in, int64_t* step_out, int64_t t_mul, int64_t t_add,
                        double lr, double b1, double b2, double eps, double gscale, int belief);
int lg_adam_plan_destroy(void* plan);
int lg_adam_epilogue_arm(const float* grad, int64_t n, const void* plan);
int lg_adam_epilogue_finish(int* applied_by_kernels, int* applied_here);
int lg_adam_epilogue_disarm(void);       /* forget everything armed without applying it (after a failed backward pass) */

/* ---- fused loss ----------------------------------------
 * loss.mse forward (loss.py:4-10) for dense fp32 tensors of n elements:
 *   err[i] = y[i] + (-y_hat[i]);   loss[0] = (sum_i err[i]^2 * (1/n)) * 0.5
 * `err` is what mse.backward multiplies by the upstream gradient (loss.py:11-12). */
int lg_mse_f32(const float* y, const float* y_hat, float* err, float* loss, int64_t n);

/* ---- integer-array indexing along one axis (csrc/index.hip) ------------------------
 * The dense tensor is seen as [outer][axis_len][inner]; `idx` holds n_idx int16/int32/int64 positions on the
 * axis (negative = from the end).  They do on the device what the reference's CPU backend does with numpy
 * (cpu/ops.py:234-255) for Dataset shuffling (data.py:15-21), embedding lookups (examples/bert.py:19-21) and
 * the label pick / update of loss.cross_entropy (loss.py:19, :22); the reference's OpenCL backend has none.
 *   lg_take_axis             dst[outer][n_idx][inner]: dst[o][j][i] = src[o][idx[j]][i]         any itemsize
 *   lg_put_axis              dst[o][idx[j]][i] = val[o][j][i]  (val == NULL: the scalar whose bits are given)
 *   lg_scatter_add_axis_f32  dst[o][idx[j]][i] += src[o][j][i]  (fp32 atomics: repeated indices accumulate;
 *                            their order of addition is not fixed, so sums of 3+ contributions may differ in
 *                            the last bit from run to run - the one non-reproducible kernel of the library)
 * pair_period = n > 0: the paired form `t[range(n), idx]` - idx has n entries, entry (o mod n) belongs to
 * outer position o, and the indexed array is [outer][inner] (no n_idx axis).
 * An index outside [-axis_len, axis_len): take writes all-ones bytes, put / scatter skip it, and the device
 * status flag is raised - the next lg_sync / lg_memcpy_d2h returns LG_EINDEX (numpy raises IndexError at
 * once; a kernel cannot). */
int lg_take_axis(int itemsize, const void* src, int64_t outer, int64_t axis_len, int64_t inner,
                 const void* idx, int idx_itemsize, int64_t n_idx, int64_t pair_period, void* dst);
int lg_put_axis(int itemsize, void* dst, int64_t outer, int64_t axis_len, int64_t inner,
                const void* idx, int idx_itemsize, int64_t n_idx, int64_t pair_period, const void* val, uint64_t scalar_bits);
int lg_scatter_add_axis_f32(float* dst, int64_t outer, int64_t axis_len, int64_t inner,
                            const void* idx, int idx_itemsize, int64_t n_idx, int64_t pair_period, const float* src);

/* Several index arrays / plain integers of one subscript folded into ONE flat index on the device (round 4): out[e] = the
 * row-major position of (idx_0[e], ..., idx_{k-1}[e]) in axes of lengths lens[0..k), e running over the arrays' common broadcast
 * `shape` (idx_strides[j][d] = stride of array j along broadcast dimension d in elements, 0 where it is broadcast; idx[j] NULL:
 * the plain integer constants[j]).  Negative indices count from the end; one out of range raises the device status flag like
 * the kernels above.  lg_mask_nonzero: the flat positions of the non-zero bytes of a boolean mask of n elements, ascending
 * (room for n), and their number - the one value a caller has to read back (it is the length of the result). */
int lg_index_fold(int k, const int64_t* lens, const void* const* idx, const int* idx_itemsize, const int64_t* constants,
                  int ndim, const int64_t* shape, const int64_t* const* idx_strides, int64_t* out);
int lg_mask_nonzero(const void* mask, int64_t n, int64_t* out_positions, int64_t* out_count);

/* ---- self-attention of a short sequence, one launch each way (csrc/attention.hip) ------------------------
 * Per (batch, head): P = softmax((Q K^T) * scale), O = P V (reference examples/bert.py:78-88: scores GEMM, `/ sqrt(d)`,
 * softmax, context GEMM - four kernels through kernels.dot / kernels.atom / the softmax composite), and its backward
 * dV = P^T dO, dS = P o (dO V^T - shift) * scale, dQ = dS K, dK = dS^T Q (dot.backward twice and the softmax composite's
 * backward).  Q, K, V, O, dO, dQ, dK, dV are addressed as (batch, position, head, d): element at X + b * sbX + pos * ldX +
 * head * D + d - the (b, s, heads * D) output of a Linear as it stands, no head-split copies; P is a dense
 * (batch, heads, S, S) tensor, written by the forward (the model returns it) and read by the backward.
 * Supported: D = 32 or 64, S = 32, 64, 96 or 128 (lg_attention_supported); operands 16-byte aligned, pitches multiples of 4.
 * The scaled scores are rounded to fp32 before the softmax and the backward's row shift is formed in double, like the separate
 * kernels (lg_softmax_scaled_f32 / _bwd_f32); sums run in a different order than the GEMM kernels', so values agree with the
 * composite form to rounding, not bit for bit. */
int lg_attention_supported(int64_t S, int64_t D);
int lg_attention_fwd_f32(const float* q, int64_t ldq, int64_t sbq, const float* k, int64_t ldk, int64_t sbk,
                         const float* v, int64_t ldv, int64_t sbv, float* o, int64_t ldo, int64_t sbo, float* p,
                         int64_t batch, int64_t heads, int64_t S, int64_t D, float scale);
int lg_attention_bwd_f32(const float* q, int64_t ldq, int64_t sbq, const float* k, int64_t ldk, int64_t sbk,
                         const float* v, int64_t ldv, int64_t sbv, const float* g, int64_t ldg, int64_t sbg,
                         const float* p, float* dq, int64_t lddq, int64_t sbdq, float* dk, int64_t lddk, int64_t sbdk,
                         float* dv, int64_t lddv, int64_t sbdv, int64_t batch, int64_t heads, int64_t S, int64_t D,
                         float scale);

/* ---- two independent products in one launch ----------------------------------------------------------
 * lg_gemm_pair_begin(); <product 1>; <product 2>; lg_gemm_pair_end();   with products issued through
 * lg_gemm_f32 / lg_gemm_rowsum_f32 / lg_gemm_fused_f32.  If the first resolves to the 64x64 tile with an
 * M-contiguous A and N-contiguous B (dW = g^T @ x of nn.Linear's backward) and the second to the 64x64 tile
 * with a K-contiguous A and N-contiguous B (dx = g @ W), neither is launched until _end, which launches both
 * as ONE grid: their workgroups share the CUs and hide each other's waits, and one launch floor disappears
 * (dot.backward of the reference is two kernels.dot calls: opencl/ops.py:127-132).  Any other product inside
 * the bracket runs as usual, after whatever was pending - results never depend on the bracket.  The two
 * products must not read each other's outputs. */
int lg_gemm_pair_begin(void);
int lg_gemm_pair_end(void);
/* THREE products in one launch: two of the first kind followed by one of the second - the backward pass of
 * `Linear -> relu -> skinny Linear -> mse` once the hidden layer's gradient exists: dW2 (+ db2) = err^T @ relu(pre) of the skinny
 * output layer (a skinny first-kind product takes the 64x64 tile inside a collecting bracket), then the hidden layer's dW1 (+ db1)
 * and dx.  The bracket may stay open between the calls (the tape runs relu.backward in between); whatever makes results visible
 * to the host or to a graph (lg_sync, device-to-host copies, graph launch, the end of a capture) launches what has been
 * collected so far, as single launches / a pair, and the bracket goes on collecting.
 * lg_gemm_pair_mse_loss: the scalar loss of lg_head_fwd_f32 ((sum(row_loss) * (1/n)) * 0.5, the value of lg_mse_finalize_f32,
 * bit for bit) rides as one spare workgroup of that three-product launch; if the bracket ends without one, it is finished by
 * a launch of its own at lg_gemm_pair_end.  row_loss and loss must stay valid until then. */
int lg_gemm_pair_mse_loss(const float* row_loss, int64_t rows, int64_t n, float* loss);
/* lg_gemm_pair_hold: the bracket stops collecting without launching what it has - products issued now run as if there were no
 * bracket (their results exist when the call returns, as always) - until lg_gemm_pair_resume makes it collect again.  The
 * caller vouches that nothing issued in between reads or writes what the held products write. */
int lg_gemm_pair_hold(void);
int lg_gemm_pair_resume(void);

/* Many weight-gradient products in ONE launch.  Between lg_gemm_group_begin and lg_gemm_group_end, lg_gemm_f32 /
 * lg_gemm_rowsum_f32 calls of the form dW (+ db) = g^T @ x (transA = 1, transB = 0, one matrix, at most 1024 output tiles of
 * 64 x 64) are prepared and QUEUED instead of launched (up to 14), and so are lg_layernorm_param_grads_f32 calls (up to 8) and
 * lg_scatter_add_rows_f32 calls of at most 4096 ids (up to 4; one more launch for these two kinds together);
 * anything else inside the bracket runs as usual.  The queue outlives the bracket: lg_gemm_group_flush launches it - ONE
 * kernel: the products, and behind them in the same grid the LayerNorm / scatter jobs (a launch of their own only when one of
 * them writes where a product writes: call order is kept then).  It is also launched when full, when a call writes where a queued one writes, and
 * before lg_sync, lg_memcpy_d2h, lg_graph_launch and the end of a capture.  Values are those of the immediate launches up to
 * rounding (a queued product always takes the 64 x 64 tile; launched at once it might split K differently); what changes is
 * WHEN they are computed: the caller must keep the queued calls' operands alive and unchanged and must not read
 * their outputs until the flush.  Made for the backward pass of a deep network: each Linear's weight gradient is a small
 * product with a long K (12 us alone, most of it launch, prologue and split-K hand-off on a handful of workgroups) that
 * nothing but the optimizer waits for; together they cost what the largest costs.  New design, no reference analog. */
int lg_gemm_group_begin(void);
/* out[c] (+)= sum over rows of in[r * ld + c]: the bias gradient of a Linear.  Inside a group bracket (one such job per
 * flush) it is queued and computed by extra workgroups of the group's launch - memory-bound work next to MFMA-bound work;
 * otherwise, or when the slot is taken, it is lg_reduce_acc at once.  Same caller obligations as for queued products. */
int lg_gemm_group_colsum_f32(const float* in, int64_t ld, int64_t rows, int64_t cols, float* out, int accumulate);
int lg_gemm_group_flush(void);
int lg_gemm_group_end(void);

/* ---- the skinny output layer and its loss (csrc/head.hip) ------------------------
 * An nn.Linear with at most 16 output features (a classifier head; reference nn.py:90-96) followed by
 * loss.mse (loss.py:4-12), forward and backward, in two launches.  `relu` != 0: `x` is the PRE-activation of
 * a relu whose output was never materialised (np.maximum(x, 0) is applied on the fly, cpu/ops.py:226).
 *   lg_head_fwd_f32   y = act(x) @ w^T + bias;  err = y + (-target);  row_loss[r] = sum_j err[r][j]^2
 *                     x: [rows, hidden] with row pitch ldx (hidden, ldx multiples of 4; x, w 16-byte aligned;
 *                     outs*hidden*4 <= 64 KiB), w: [outs, hidden] dense, bias: [outs] or NULL, target/y/err:
 *                     [rows, outs] dense, row_loss: [rows].
 *   lg_mse_finalize_f32   loss[0] = (sum_r row_loss[r] * (1/n)) * 0.5 with n = rows*outs (loss.py:9-10): one small
 *                     launch, needed only when the loss is read before lg_head_bwd_f32 has run (which finishes it
 *                     with one extra workgroup, in the same summation order - same bits either way).
 *   lg_head_bwd_f32   dx = g @ w  ([rows, hidden] dense, or NULL);  gpre = dx * (x >= 0)  (relu.backward's result,
 *                     cpu/ops.py:229; or NULL; needs relu != 0);  dw (+)= g^T @ act(x)  ([outs, hidden] dense, or
 *                     NULL);  db (+)= column sums of g  ([outs], or NULL)  - what dot.backward, transpose.backward
 *                     and the bias un-broadcast of func.py:50-56 compute for `act(x) @ w.T(1,0) + b`;
 *                     row_loss / loss: both NULL, or the row sums of lg_head_fwd_f32 and where to put the scalar.
 * They replace pyopencl launches of the reference's OpenCL backend: kernels.dot (opencl/ops.py:119-132),
 * kernels.atom (:285-288) and kernels.reduce (:344-368). */
int lg_head_fwd_f32(const float* x, int64_t ldx, int relu, const float* w, const float* bias, const float* target,
                    float* y, float* err, float* row_loss, int64_t rows, int64_t hidden, int64_t outs);
/* lg_head_fwd_f32 that also writes dx[rows, hidden] = err @ w and gpre[rows, hidden] = dx * (x >= 0) (relu != 0): what
 * lg_head_bwd_f32 writes for g = err, bit for bit - the gradients of the loss with respect to the layer's input and to the
 * pre-activation when backward() starts at this loss (its seed is 1: loss.py:12 hands `err` on as the output layer's gradient).
 * Computed while the row of x is in the cache and w in LDS; the caller uses them in the backward pass only if that is how the
 * pass goes and x, w, err are unchanged by then.  dx and gpre: both NULL (= lg_head_fwd_f32) or both given. */
int lg_head_fwd_grad_f32(const float* x, int64_t ldx, int relu, const float* w, const float* bias, const float* target,
                         float* y, float* err, float* row_loss, float* dx, float* gpre, int64_t rows, int64_t hidden, int64_t outs);
/* The hidden layer in front of such a head and the head itself: pre = x @ w1^T + b1 ([rows, hidden] dense; x: [rows, d_in] with row
 * pitch ldx, w1: [hidden, d_in] with row pitch ldw1 - nn.Linear's layout), then lg_head_fwd_grad_f32 on pre: two launches.
 * chain != 0: ONE launch when the product resolves to the 64x32 two-K-group tile (the MNIST MLP's 1024 x 512 x 784 does),
 * hidden <= 1024 and rows <= 4096 - the head's rows are computed by workgroups at the end of the product's grid, each waiting
 * for the tiles of its rows (written through, announced by a flag).  Same bits either way.  An experiment: measured SLOWER than
 * the two launches on MI355X (22.0 against 20.6 us, profiles/r4/chain_bench.txt) - the tape calls the two launches.
 * *launches (may be NULL) receives 1 or 2. */
int lg_gemm_bias_head_fwd_f32(const float* x, int64_t ldx, const float* w1, int64_t ldw1, const float* b1, float* pre,
                              int64_t rows, int64_t hidden, int64_t d_in, int relu,
                              const float* w2, const float* b2, const float* target, float* y, float* err, float* row_loss,
                              float* dx, float* gpre, int64_t outs, int chain, int* launches);
int lg_mse_finalize_f32(const float* row_loss, int64_t rows, int64_t n, float* loss);
int lg_head_bwd_f32(const float* x, int64_t ldx, int relu, const float* g, const float* w,
                    float* dx, float* gpre, float* dw, int dw_accumulate, float* db, int db_accumulate,
                    int64_t rows, int64_t hidden, int64_t outs, const float* row_loss, float* loss);

/* ---- row-wise fused ops for the tiny-BERT path -------
 * Dense fp32 [rows, cols] operands; one wavefront owns a row.
 * softmax over the last axis = the composite of autograd/ops.py:62-66; backward dx = y * (g - sum(g*y)).
 * layernorm = the composite of nn.py:109-124 for a 1-D normalised shape; saves xhat and rstd for the backward,
 * which returns dx only (dw = sum_rows(g * xhat), db = sum_rows(g) are lg_ew + lg_reduce calls). */
int lg_softmax_f32(const float* x, float* y, int64_t rows, int64_t cols);
int lg_softmax_bwd_f32(const float* y, const float* g, float* dx, int64_t rows, int64_t cols);
/* softmax(x * scale) and its backward (dx of the UNSCALED x) - attention's `(q @ k / sqrt(d)).softmax(-1)` (reference
 * examples/bert.py:81-86) without the separate scaling passes; x * scale is rounded to fp32 before anything else, and the
 * backward multiplies the finished fp32 gradient, so both give the bits of the two-kernel form. */
int lg_softmax_scaled_f32(const float* x, float* y, int64_t rows, int64_t cols, float scale);
int lg_softmax_scaled_bwd_f32(const float* y, const float* g, float* dx, int64_t rows, int64_t cols, float scale);
int lg_layernorm_f32(const float* x, const float* w, const float* b, float* y, float* xhat, float* rstd,
                     int64_t rows, int64_t cols, double eps);
int lg_layernorm_bwd_f32(const float* g, const float* w, const float* xhat, const float* rstd, float* dx,
                         int64_t rows, int64_t cols);
/* parameter gradients of the same LayerNorm from one launch: dw[c] (+)= sum_r g[r][c] * xhat[r][c], db[c] (+)= sum_r g[r][c]
 * (the tape's mul + two un-broadcasting column sums of func.py:50-56 + two `grad +=`) */
int lg_layernorm_param_grads_f32(const float* g, const float* xhat, float* dw, float* db, int64_t rows, int64_t cols,
                                 int dw_accumulate, int db_accumulate);
/* embedding lookup out[i, :] = table[ids[i], :] (ids int32 or int64, negative ids wrap like numpy) and its
 * gradient grad_table[ids[i], :] += grad_out[i, :] (float atomics: repeated ids ACCUMULATE; the reference's
 * numpy `grad[idx] = g`, cpu/ops.py:245, keeps only the last one, and its BERT example drops the gradient). */
int lg_gather_rows_f32(const float* table, const void* ids, int id_itemsize, float* out,
                       int64_t n_ids, int64_t row_len, int64_t table_rows);
/* out[i, :] = (t0[ids0[i % n0], :] + t1[ids1[i % n1], :]) + t2[ids2[i % n2], :] - three lookups and their sum in one pass
 * (BERT's word + position + token-type embeddings, examples/bert.py:36-40: three lookups and two kernels.atom adds).  n_k ids
 * for table k (n_k divides n_out: an id tensor broadcast over leading axes), rows_k rows in table k; ids all int32 or all
 * int64.  Sums in the order written, each rounded to fp32.  Out-of-range ids as in lg_gather_rows_f32. */
int lg_gather_sum3_rows_f32(const float* t0, const void* ids0, int64_t n0, int64_t rows0,
                            const float* t1, const void* ids1, int64_t n1, int64_t rows1,
                            const float* t2, const void* ids2, int64_t n2, int64_t rows2,
                            int id_itemsize, float* out, int64_t n_out, int64_t row_len);
int lg_scatter_add_rows_f32(const float* grad_out, const void* ids, int id_itemsize, float* grad_table,
                            int64_t n_ids, int64_t row_len, int64_t table_rows);

/* loss.cross_entropy (loss.py:14-24) for dense fp32 logits [rows, cols] and integer labels [rows] (int16/32/64):
 *   nll[r] = -log(softmax(logits[r])[label[r]]);  dlogits[r][c] = (softmax(logits[r])[c] - [c == label[r]]) / rows
 * the loss is mean(nll) (lg_reduce + one scalar multiply), its gradient dlogits * upstream. */
int lg_cross_entropy_f32(const float* logits, const void* labels, int label_itemsize, float* dlogits, float* nll,
                         int64_t rows, int64_t cols);
/* The same plus the loss itself: mean[0] = (sum of nll) * (1 / rows), as `loss.cross_entropy` returns it (loss.py:21).  For
 * vocabulary-sized rows the mean is formed inside the row kernel (the workgroup that finishes last sums the row losses in a
 * fixed order); otherwise by the generic reduction + scaling behind it. */
int lg_cross_entropy_mean_f32(const float* logits, const void* labels, int label_itemsize, float* dlogits, float* nll,
                              float* mean, int64_t rows, int64_t cols);

/* library build info: "liblghip <version> gfx950 <build date>" */
const char* lg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LGHIP_H */
