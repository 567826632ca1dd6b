// Context, RNG initialisation, cuckoo membership, pi initialisation, neighbour sampler and the
// wg_* primitive test kernels.  Written for gfx950 (wave64) only.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ammsb_ctx.h"
#include "ammsb_dev.h"
#include "ammsb_step.h"

using namespace ammsb;

// ------------------------------------------------------------------------------ misc / ctx

extern "C" int ammsb_version(void) { return AMMSB_VERSION; }

extern "C" const char* ammsb_strerror(int code) {
  switch (code) {
    case AMMSB_OK: return "ok";
    case AMMSB_EINVAL: return "invalid argument";
    case AMMSB_EHIP: return "HIP runtime error";
    case AMMSB_ENOMEM: return "out of device memory";
    case AMMSB_ENODEV: return "no usable gfx950 device";
    case AMMSB_ERANGE: return "size out of range for the launch shape";
    default: return "unknown error";
  }
}

extern "C" const char* ammsb_last_error(const ammsb_ctx* ctx) { return ctx ? ctx->err : "no context"; }

extern "C" const char* ammsb_last_kernel_name(const ammsb_ctx* ctx, int which) {
  if (!ctx || which < 0 || which > 4 || !ctx->kernel_name[which]) return "";
  return ctx->kernel_name[which];
}

// config.cc:57-64 float_to_string: "%e" (6 digits) + 'f', re-read by the kernel compiler
static float quantize(float v) {
  char buf[64];
  snprintf(buf, sizeof buf, "%e", (double)v);
  return strtof(buf, nullptr);
}

extern "C" int ammsb_params_quantize(ammsb_params* p) {
  if (!p) return AMMSB_EINVAL;
  p->alpha = quantize(p->alpha);
  p->a = quantize(p->a);
  p->b = quantize(p->b);
  p->c = quantize(p->c);
  p->epsilon = quantize(p->epsilon);
  p->eta0 = quantize(p->eta0);
  p->eta1 = quantize(p->eta1);
  return AMMSB_OK;
}

// learner.cc:41-43: EPS_A * pow(1 + step_count / EPS_B, -EPS_C), pow in binary64 rounded once
extern "C" float ammsb_eps_t(const ammsb_params* p, uint32_t step_count) {
  const float t = 1.0f + (float)step_count / p->b;
  return p->a * (float)pow((double)t, (double)-p->c);
}

extern "C" int ammsb_ctx_create(int device_id, const ammsb_params* params, ammsb_ctx** out) {
  if (!params || !out) return AMMSB_EINVAL;
  if (params->K == 0 || params->N == 0 || params->num_node_sample == 0) return AMMSB_EINVAL;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return AMMSB_ENODEV;
  if (device_id < 0 || device_id >= count) return AMMSB_ENODEV;
  if (hipSetDevice(device_id) != hipSuccess) return AMMSB_EHIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return AMMSB_EHIP;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return AMMSB_ENODEV;  // the code object is gfx950-only
  ammsb_ctx* ctx = static_cast<ammsb_ctx*>(calloc(1, sizeof(ammsb_ctx)));
  if (!ctx) return AMMSB_ENOMEM;
  ctx->device = device_id;
  ctx->params = *params;
  ctx->num_cus = prop.multiProcessorCount;
  ctx->max_partials = 4096;
  ctx->max_ppx_blocks = 4096;
  const size_t K = params->K;
  if (hipMalloc(&ctx->grad_partials, sizeof(float) * ctx->max_partials * 2 * K) != hipSuccess ||
      hipMalloc(&ctx->theta_sum, sizeof(float) * K) != hipSuccess ||
      hipMalloc(&ctx->theta_coef, sizeof(float4) * K) != hipSuccess ||
      hipMalloc(&ctx->ppx_partials, sizeof(double) * ctx->max_ppx_blocks * 2) != hipSuccess ||
      hipMalloc(&ctx->ppx_cnt_partials, sizeof(unsigned long long) * ctx->max_ppx_blocks * 2) != hipSuccess ||
      hipMalloc(&ctx->ppx_ticket, sizeof(uint32_t)) != hipSuccess ||
      hipMemset(ctx->ppx_ticket, 0, sizeof(uint32_t)) != hipSuccess) {
    ammsb_ctx_destroy(ctx);
    return AMMSB_ENOMEM;
  }
  *out = ctx;
  return AMMSB_OK;
}

extern "C" int ammsb_ctx_destroy(ammsb_ctx* ctx) {
  if (!ctx) return AMMSB_EINVAL;
  (void)hipSetDevice(ctx->device);
  if (ctx->grad_partials) (void)hipFree(ctx->grad_partials);
  if (ctx->theta_sum) (void)hipFree(ctx->theta_sum);
  if (ctx->theta_coef) (void)hipFree(ctx->theta_coef);
  if (ctx->ppx_partials) (void)hipFree(ctx->ppx_partials);
  if (ctx->ppx_cnt_partials) (void)hipFree(ctx->ppx_cnt_partials);
  if (ctx->ppx_ticket) (void)hipFree(ctx->ppx_ticket);
  free(ctx);
  return AMMSB_OK;
}

extern "C" int ammsb_ctx_params(const ammsb_ctx* ctx, ammsb_params* out) {
  if (!ctx || !out) return AMMSB_EINVAL;
  *out = ctx->params;
  return AMMSB_OK;
}

static inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// ----------------------------------------------------------------------------------- RNG init

__global__ void rng_init_kernel(ammsb_seed* seeds, uint64_t n, uint64_t sx, uint64_t sy) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    seeds[i].x = sx + i;  // random.cc:36-39
    seeds[i].y = sy + i;
  }
}

extern "C" int ammsb_rng_init(ammsb_ctx* ctx, ammsb_seed* seeds, uint64_t n, uint64_t sx, uint64_t sy, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && seeds, "null argument");
  if (n == 0) return AMMSB_OK;
  const uint32_t grid = div_up(n, 256) < 4096 ? div_up(n, 256) : 4096;
  rng_init_kernel<<<grid, 256, 0, as_stream(stream)>>>(seeds, n, sx, sy);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

// Streams for NEW consumers (the device mini-batch sampler): the reference's {sx+i, sy+i} layout gives
// neighbouring streams almost identical small states, and the first outputs of xorshift128+ from such states are
// strongly correlated across streams (measured: 75k second draws mod 1M gave only 63.8k distinct values, where
// independent draws give 72.5k).  Each word is passed through the SplitMix64 finaliser instead.
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__global__ void rng_init_mixed_kernel(ammsb_seed* seeds, uint64_t n, uint64_t sx, uint64_t sy) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = splitmix64(sx + 2 * i), y = splitmix64(sy + 2 * i + 1);
  if ((x | y) == 0) x = 1;  // the all-zero state is xorshift128+'s fixed point
  seeds[i].x = x;
  seeds[i].y = y;
}

extern "C" int ammsb_rng_init_mixed(ammsb_ctx* ctx, ammsb_seed* seeds, uint64_t n, uint64_t sx, uint64_t sy,
                                    void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && seeds, "null argument");
  if (n == 0) return AMMSB_OK;
  const uint64_t grid = (n + 255) / 256;
  AMMSB_CHECK_ARG(ctx, grid <= 0x7fffffffull, "too many streams");
  rng_init_mixed_kernel<<<(uint32_t)grid, 256, 0, as_stream(stream)>>>(seeds, n, sx, sy);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

__global__ void randn_fill_kernel(ammsb_seed* seeds, uint32_t n_streams, uint32_t per_stream, float* out) {
  __shared__ ZigTables zig;
  zig_load(&zig);
  __syncthreads();
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_streams) return;
  ammsb_seed s = seeds[gid];
  float* o = out + (uint64_t)gid * per_stream;
  for (uint32_t i = 0; i < per_stream; ++i) o[i] = rng_normal(s, &zig);
  seeds[gid] = s;
}

extern "C" int ammsb_randn_fill(ammsb_ctx* ctx, ammsb_seed* seeds, uint32_t n_streams, uint32_t per_stream,
                                float* out, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && seeds && out, "null argument");
  if (n_streams == 0) return AMMSB_OK;
  randn_fill_kernel<<<div_up(n_streams, 64), 64, 0, as_stream(stream)>>>(seeds, n_streams, per_stream, out);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

// ------------------------------------------------------------------------------------ cuckoo

__global__ void set_has_kernel(ammsb_set set, const uint64_t* keys, uint64_t n, uint8_t* out) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = set_has(set, keys[i]) ? 1 : 0;
}

extern "C" int ammsb_set_has(ammsb_ctx* ctx, const ammsb_set* set, const uint64_t* keys, uint64_t n, uint8_t* out,
                             void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && set && set->slots && keys && out, "null argument");
  AMMSB_CHECK_ARG(ctx, set->num_bins > 0 && set->prime_idx < 4, "bad set descriptor");
  if (n == 0) return AMMSB_OK;
  const uint32_t grid = div_up(n, 256) < 8192 ? div_up(n, 256) : 8192;
  set_has_kernel<<<grid, 256, 0, as_stream(stream)>>>(*set, keys, n, out);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

static int check_rpm(ammsb_ctx* ctx, const ammsb_rpm* m) {
  AMMSB_CHECK_ARG(ctx, m, "null matrix");
  AMMSB_CHECK_ARG(ctx, m->num_blocks >= 1 && m->num_blocks <= AMMSB_RPM_MAX_BLOCKS, "bad block count");
  AMMSB_CHECK_ARG(ctx, m->rows_in_block > 0 && m->num_cols > 0, "bad block shape");
  AMMSB_CHECK_ARG(ctx, (uint64_t)m->num_blocks * m->rows_in_block >= m->num_rows, "blocks do not cover rows");
  for (uint32_t i = 0; i < m->num_blocks; ++i) AMMSB_CHECK_ARG(ctx, m->blocks[i], "null block");
  return AMMSB_OK;
}

// ----------------------------------------------------------------------------------- pi init
// generate_gamma (random.cc:108-126) fused with WG_NORMALIZE_PARTITIONED_KERNEL (normalize.cc:34-52),
// both launched by the reference with 32-lane groups: lane l of group g fills columns l, l+32, ...
// of rows g, g+G, ... from stream g*32+l; its WG_SUM partial is over exactly those columns, so the
// row sum is accumulated while generating.

__global__ __launch_bounds__(64) void pi_init_kernel(ammsb_rpm pi, float* phi_sum, float eta0, float eta1,
                                                      ammsb_seed* seeds, uint32_t G) {
  using Grp = Group<32>;
  __shared__ ZigTables zig;
  zig_load(&zig);
  __syncthreads();
  const uint32_t g = blockIdx.x * Grp::PER_BLOCK + Grp::slot();
  const uint32_t l = Grp::lane();
  const bool live = g < G;
  ammsb_seed s = {0, 0};
  if (live) s = seeds[(uint64_t)g * 32 + l];
  const uint64_t N = pi.num_rows, K = pi.num_cols;
  const uint64_t trips = (N + G - 1) / G;
  int phase = 0;
  for (uint64_t t = 0; t < trips; ++t) {
    const uint64_t row = g + t * G;
    const bool on = live && row < N;
    float* r = on ? rpm_row(pi, row) : nullptr;
    float lsum = 0;
    if (on)
      for (uint64_t j = l; j < K; j += 32) {
        const float v = rng_gamma(s, &zig, eta0, eta1);
        r[j] = v;
        lsum += v;
      }
    const float sum = Grp::sum(lsum, (float*)nullptr, phase);
    if (on) {
      for (uint64_t j = l; j < K; j += 32) r[j] = r[j] / sum;
      if (l == 0) phi_sum[row] = sum;
    }
  }
  if (live) seeds[(uint64_t)g * 32 + l] = s;
}

extern "C" int ammsb_pi_init_gamma(ammsb_ctx* ctx, const ammsb_rpm* pi, float* phi_sum, float eta0, float eta1,
                                   ammsb_seed* seeds, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && phi_sum && seeds, "null argument");
  int rc = check_rpm(ctx, pi);
  if (rc) return rc;
  const uint32_t G = pi->num_rows < AMMSB_MAX_GROUPS ? (uint32_t)pi->num_rows : AMMSB_MAX_GROUPS;  // random.cc:154
  pi_init_kernel<<<div_up(G, 2), 64, 0, as_stream(stream)>>>(*pi, phi_sum, eta0, eta1, seeds, G);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

// ------------------------------------------------------------------------- neighbour sampler
// generate_random_int_kernel, sample.cc:13-78.  Virtual thread gid owns stream gid and nodes
// gid, gid+gsize, ...; the open-addressing table (capacity 2n, empty marker N) lives in the
// caller's `table` buffer exactly as in the reference (NeighborSampler::GetHash()).

__device__ __forceinline__ void ns_generate(ammsb_seed& seed, uint32_t* out, uint32_t capacity, uint32_t max_id,
                                            uint32_t node) {
  uint32_t r, val;
  do {
    do {
      r = (uint32_t)(rng_next(seed) % (uint64_t)(max_id + 1));  // randint(seed, 0, max_id), random.cl.inc:37-39
    } while (r == node);
    const uint32_t l1 = (r ^ 553105253u) % capacity;
    const uint32_t l2 = 1u + (capacity << 1);
    for (uint32_t i = 0;; ++i) {
      const uint32_t offset = (l1 + i * l2) % capacity;
      val = out[offset];
      if (val == r) break;
      if (val == max_id + 1) {
        out[offset] = r;
        break;
      }
    }
  } while (val == r);
}

// sample.cc:116-119: the reference's global size for n_nodes samples and work-group size wg
__host__ __device__ __forceinline__ uint32_t ns_global_size(uint32_t n_nodes, uint32_t wg) {
  uint32_t groups = n_nodes / wg + (n_nodes % wg ? 1 : 0);
  const uint32_t maxg = AMMSB_MAX_GROUPS / wg;
  if (groups > maxg) groups = maxg;
  return groups * wg;
}

__global__ __launch_bounds__(64) void sample_neighbors_kernel(ammsb_seed* seeds, const uint32_t* nodes,
                                                               uint32_t n_nodes, uint32_t N, uint32_t n,
                                                               uint32_t gsize, uint32_t* table, uint32_t* packed,
                                                               const ammsb_step_desc* desc, uint32_t wg) {
  if (desc) {  // captured graph: the grid covers the largest mini-batch
    n_nodes = desc->n_nodes;
    gsize = ns_global_size(n_nodes, wg);
  }
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= gsize || gid >= n_nodes) return;
  const uint32_t capacity = 2 * n;
  ammsb_seed seed = seeds[gid];
  for (uint32_t i = gid; i < n_nodes; i += gsize) {
    uint32_t* out = table + (uint64_t)i * capacity;
    uint32_t* pk = packed + (uint64_t)i * n;
    const uint32_t node = nodes[i];
    for (uint32_t j = 0; j < capacity; ++j) out[j] = N;
    for (uint32_t j = 0; j < n; ++j) ns_generate(seed, out, capacity, N - 1, node);
    uint32_t count = 0;
    for (uint32_t j = 0; j < capacity && count < n; ++j) {
      const uint32_t v = out[j];
      if (v != N) pk[count++] = v;
    }
  }
  seeds[gid] = seed;
}

// Same algorithm with each lane's open-addressing table held in LDS (column `lane` of a
// [capacity][64] array: one bank per lane whatever slot it probes).  The global-memory form above
// chains ~2n dependent round trips to HBM per node; here the only global traffic is the node id, the
// stream state and the two coalesced result arrays (the table image is still written out:
// NeighborSampler::GetHash() exposes it).  Results are identical by construction.
__global__ __launch_bounds__(64) void sample_neighbors_lds_kernel(ammsb_seed* seeds, const uint32_t* nodes,
                                                                   uint32_t n_nodes, uint32_t N, uint32_t n,
                                                                   uint32_t gsize, uint32_t* table, uint32_t* packed,
                                                                   const ammsb_step_desc* desc, uint32_t wg) {
  if (desc) {  // captured graph: the grid covers the largest mini-batch
    n_nodes = desc->n_nodes;
    gsize = ns_global_size(n_nodes, wg);
    if (blockIdx.x * 64 >= gsize || blockIdx.x * 64 >= n_nodes) return;  // block-uniform: no lane owns a node
  }
  // [capacity][65]: slot j of lane l at j * 65 + l.  A lane probing its own table touches one bank whatever the
  // slot; the transposed read-back below (consecutive lanes = consecutive slots of one node) strides by 65 and
  // is conflict-free too (a stride of 64 would put all 64 lanes on one bank).
  extern __shared__ uint32_t lds[];
  constexpr uint32_t S = 65;
  const uint32_t lane = threadIdx.x;
  const uint32_t gid = blockIdx.x * 64 + lane;
  const uint32_t capacity = 2 * n;
  const bool owner = gid < gsize && gid < n_nodes;
  ammsb_seed seed = owner ? seeds[gid] : ammsb_seed{0, 0};
  const uint32_t max_id = N - 1;
  const FastMod mod_n = fast_mod_init((uint64_t)max_id + 1);
  const uint64_t cap_m = ~0ull / capacity + 1;  // Lemire: a % capacity == mulhi64(cap_m * a, capacity) for 32-bit a
  // (row, col) of element x = lane + 64 k of a row-major [live][width] array, advanced without a division
  auto advance = [](uint32_t& row, uint32_t& col, uint32_t width) {
    col += 64;
    while (col >= width) {
      col -= width;
      ++row;
    }
  };
  for (uint32_t base = blockIdx.x * 64; base < n_nodes; base += gsize) {  // uniform across the block
    const uint32_t i = base + lane;
    const bool on = owner && i < n_nodes;
    const uint32_t live = min(64u, min(n_nodes - base, gsize - blockIdx.x * 64));  // lanes with a node this round
    for (uint32_t j = 0; j < capacity; ++j) lds[j * S + lane] = N;
    if (on) {
      const uint32_t node = nodes[i];
      for (uint32_t j = 0; j < n; ++j) {
        uint32_t r, val;
        do {
          do {
            r = (uint32_t)fast_mod(rng_next(seed), mod_n);  // randint(seed, 0, max_id), random.cl.inc:37-39
          } while (r == node);
          // h1 = (r ^ 553105253) % capacity; the probe step 1 + 2 capacity is 1 modulo capacity (sample.cc:15-21)
          uint32_t offset = (uint32_t)__umul64hi(cap_m * (uint64_t)(r ^ 553105253u), capacity);
          for (;;) {
            val = lds[offset * S + lane];
            if (val == r) break;
            if (val == max_id + 1) {
              lds[offset * S + lane] = r;
              break;
            }
            offset = offset + 1 == capacity ? 0 : offset + 1;
          }
        } while (val == r);
      }
    }
    __syncthreads();
    // table image: `live` consecutive rows of `capacity` words
    uint32_t* tout = table + (uint64_t)base * capacity;
    {
      uint32_t row = 0, col = lane;
      while (col >= capacity) {
        col -= capacity;
        ++row;
      }
      for (uint32_t x = lane; x < live * capacity; x += 64) {
        tout[x] = lds[col * S + row];
        advance(row, col, capacity);
      }
    }
    __syncthreads();
    if (on) {  // compact in place: count <= j, so the write never overtakes the read
      uint32_t count = 0;
      for (uint32_t j = 0; j < capacity && count < n; ++j) {
        const uint32_t v = lds[j * S + lane];
        if (v != N) lds[(count++) * S + lane] = v;
      }
    }
    __syncthreads();
    uint32_t* pout = packed + (uint64_t)base * n;
    {
      uint32_t row = 0, col = lane;
      while (col >= n) {
        col -= n;
        ++row;
      }
      for (uint32_t x = lane; x < live * n; x += 64) {
        pout[x] = lds[col * S + row];
        advance(row, col, n);
      }
    }
    __syncthreads();
  }
  if (owner) seeds[gid] = seed;
}

// n = 32 (capacity 2n = 64 = one wave): ONE WAVE PER STREAM.  The per-thread kernels above run the reference's
// algorithm as written -- one lane per node, a dependent chain of n draws, each a 64-bit modulo, a hash and a table
// probe -- and take ~1.2 us per draw whatever the node count (latency-bound: 33 nodes 32 us, 8193 nodes 42 us), which
// made the sampler chain the longest thing in a small iteration.  Here the same sequence is produced with the work
// split by what is actually sequential:
//   1. the xorshift128+ stream: all lanes step it together (wave-uniform arithmetic), lane j keeps output j and the
//      state after it -- B raw draws per batch;
//   2. every lane reduces ITS draw modulo N and hashes it (the expensive part: once, in parallel, not per draw);
//   3. the open-addressing table IS the wave: slot s lives in lane s.  Draws are consumed in stream order; "is it
//      there already" is one ballot, "first free slot at or after h" a rotate + find-first-set on the occupancy mask.
// The table image, the packed result (table entries in slot order, sample.cc:64-76) and the stream state after the
// last consumed draw are those of the per-thread kernel bit for bit (same tests).
__global__ __launch_bounds__(256) void sample_neighbors_wave_kernel(ammsb_seed* seeds, const uint32_t* nodes,
                                                                     uint32_t n_nodes, uint32_t N, uint32_t gsize,
                                                                     uint32_t* table, uint32_t* packed,
                                                                     const ammsb_step_desc* desc, uint32_t wg) {
  constexpr uint32_t CAP = 64, NN = 32, B = 40;  // capacity, neighbours per node, raw draws per batch
  if (desc) {
    n_nodes = desc->n_nodes;
    gsize = ns_global_size(n_nodes, wg);
  }
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t gid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));  // stream = wave
  if (gid >= gsize || gid >= n_nodes) return;  // wave-uniform
  ammsb_seed st = seeds[gid];
  const FastMod mod_n = fast_mod_init((uint64_t)N);
  for (uint32_t i = gid; i < n_nodes; i += gsize) {
    const uint32_t node = nodes[i];
    uint32_t slot = N;  // lane s = table slot s; N marks an empty slot (sample.cc:57)
    uint32_t count = 0;
    while (count < NN) {
      // 1. B raw draws from the stream; lane j keeps draw j and the state after it
      ammsb_seed run = st, mine = st;
      uint64_t raw = 0;
#pragma unroll 8
      for (uint32_t j = 0; j < B; ++j) {
        const uint64_t x = rng_next(run);
        if (lane == j) {
          raw = x;
          mine = run;
        }
      }
      // 2. randint(seed, 0, N - 1) and the slot hash of this lane's draw
      const uint32_t r = (uint32_t)fast_mod(raw, mod_n);
      const uint32_t h = (r ^ 553105253u) & (CAP - 1);  // (r ^ c) % capacity; the probe step 1 + 2 capacity is 1
      // 3. consume the draws in stream order
      uint32_t used = 0;
      for (uint32_t j = 0; j < B && count < NN; ++j) {
        used = j + 1;
        const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)j);
        if (rj == node) continue;                                  // do { r = randint } while (r == node)
        if (__ballot(slot == rj) != 0ull) continue;                // already in the table: draw again
        const uint32_t hj = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)j);
        const unsigned long long freeb = ~__ballot(slot != N);     // never zero: at most 32 of 64 slots are taken
        const unsigned long long rot = hj ? ((freeb >> hj) | (freeb << (64 - hj))) : freeb;
        const uint32_t pos = (hj + (uint32_t)__builtin_ctzll(rot)) & (CAP - 1);
        if (lane == pos) slot = rj;
        ++count;
      }
      // the stream stands after the last draw consumed
      auto lane64 = [&](uint64_t v) -> uint64_t {  // (readlane returns int: cast before widening)
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)(used - 1));
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)(used - 1));
        return ((uint64_t)hi << 32) | lo;
      };
      st.x = lane64(mine.x);
      st.y = lane64(mine.y);
    }
    // table image (GetHash()) and the packed result: occupied slots in slot order
    table[(uint64_t)i * CAP + lane] = slot;
    const unsigned long long occ = __ballot(slot != N);
    if (slot != N) packed[(uint64_t)i * NN + __popcll(occ & ((1ull << lane) - 1ull))] = slot;
  }
  if (lane == 0) seeds[gid] = st;
}

static int sample_neighbors_common(ammsb_ctx* ctx, ammsb_seed* seeds, const uint32_t* nodes, uint32_t n_nodes,
                                   uint32_t wg, uint32_t* table, uint32_t* packed, const ammsb_step_desc* desc,
                                   void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && seeds && nodes && table && packed, "null argument");
  AMMSB_CHECK_ARG(ctx, wg >= 1 && wg <= 1024, "sampler wg out of range");
  if (n_nodes == 0) return AMMSB_OK;
  const uint32_t n = ctx->params.num_node_sample;
  AMMSB_CHECK_ARG(ctx, ctx->params.N > (uint64_t)n + 1, "N must exceed num_node_sample + 1");
  const uint32_t gsize = ns_global_size(n_nodes, wg);  // sample.cc:116-119
  const size_t lds_bytes = (size_t)2 * n * 65 * sizeof(uint32_t);
  static const bool per_thread = [] {  // AMMSB_NBR_FORM=t: the per-thread kernels where the wave form would be picked
    const char* f = getenv("AMMSB_NBR_FORM");
    return f && f[0] == 't';
  }();
  // one wave per stream pays while there are few streams (33 nodes: 8 vs 32 us, 8193: 23 vs 42 us); with 65537 of
  // them the chip is full either way and 64 nodes per wave use it better (72 vs 132 us)
  if (n == 32 && !per_thread && n_nodes <= 20000) {
    const uint32_t waves = gsize < n_nodes ? gsize : n_nodes;
    sample_neighbors_wave_kernel<<<div_up(waves, 4), 256, 0, as_stream(stream)>>>(
        seeds, nodes, n_nodes, (uint32_t)ctx->params.N, gsize, table, packed, desc, wg);
  } else if (lds_bytes <= 64 * 1024)
    sample_neighbors_lds_kernel<<<div_up(gsize, 64), 64, lds_bytes, as_stream(stream)>>>(
        seeds, nodes, n_nodes, (uint32_t)ctx->params.N, n, gsize, table, packed, desc, wg);
  else
    sample_neighbors_kernel<<<div_up(gsize, 64), 64, 0, as_stream(stream)>>>(
        seeds, nodes, n_nodes, (uint32_t)ctx->params.N, n, gsize, table, packed, desc, wg);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

extern "C" int ammsb_sample_neighbors(ammsb_ctx* ctx, ammsb_seed* seeds, const uint32_t* nodes, uint32_t n_nodes,
                                      uint32_t wg, uint32_t* table, uint32_t* packed, void* stream) {
  return sample_neighbors_common(ctx, seeds, nodes, n_nodes, wg, table, packed, nullptr, stream);
}

int ammsb_sample_neighbors_d(ammsb_ctx* ctx, ammsb_seed* seeds, const uint32_t* nodes, uint32_t n_nodes_cap, uint32_t wg,
                             uint32_t* table, uint32_t* packed, const ammsb_step_desc* desc, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && desc, "null descriptor");
  return sample_neighbors_common(ctx, seeds, nodes, n_nodes_cap, wg, table, packed, desc, stream);
}

// ----------------------------------------------------------------------------- wg primitives
// Literal LDS restatements of algorithm/{sum,normalize,sort}.cc for ANY work-group size (the
// reference tests use {2,4,16,32,64,96,113}); the hot kernels use Group<L> instead.

__device__ __forceinline__ uint32_t power_of_2(uint32_t v) {  // sum.cc:11-18
  v |= v >> 1;
  v |= v >> 2;
  v |= v >> 4;
  v |= v >> 8;
  v |= v >> 16;
  return v + 1;
}

template <typename T>
__device__ __forceinline__ T wg_sum_lds(const T* in, T* aux, uint32_t len) {  // sum.cc:20-42
  const uint32_t lid = threadIdx.x, lsize = blockDim.x;
  T lsum = 0;
  for (uint32_t i = lid; i < len; i += lsize) lsum += in[i];
  aux[lid] = lsum;
  __syncthreads();
  for (uint32_t p2 = power_of_2(lsize) >> 1; p2 > 0; p2 >>= 1) {
    if (lid < p2 && lid + p2 < lsize) aux[lid] += aux[lid + p2];
    __syncthreads();
  }
  const T r = aux[0];
  __syncthreads();
  return r;
}

template <typename T>
__global__ void wg_sum_kernel(const T* in, T* out, uint32_t rows, uint32_t len) {  // sum.cc:44-52
  __shared__ T aux[1024];
  for (uint32_t gid = blockIdx.x; gid < rows; gid += gridDim.x) {
    const T s = wg_sum_lds(in + (uint64_t)gid * len, aux, len);
    if (threadIdx.x == 0) out[gid] = s;
  }
}

__global__ void wg_normalize_kernel(float* in, float* sums, uint32_t rows, uint32_t len) {  // normalize.cc:13-32
  __shared__ float aux[1024];
  for (uint32_t gid = blockIdx.x; gid < rows; gid += gridDim.x) {
    float* row = in + (uint64_t)gid * len;
    const float sum = wg_sum_lds(row, aux, len);
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) row[i] = row[i] / sum;
    if (sums && threadIdx.x == 0) sums[gid] = sum;
    __syncthreads();
  }
}

__global__ void rpm_sum_kernel(ammsb_rpm m, float* out, int normalize) {  // sum.cc:54-65, normalize.cc:34-52
  __shared__ float aux[1024];
  for (uint64_t gid = blockIdx.x; gid < m.num_rows; gid += gridDim.x) {
    float* row = rpm_row(m, gid);
    const float sum = wg_sum_lds(row, aux, (uint32_t)m.num_cols);
    if (normalize)
      for (uint64_t i = threadIdx.x; i < m.num_cols; i += blockDim.x) row[i] = row[i] / sum;
    if (threadIdx.x == 0) out[gid] = sum;
    __syncthreads();
  }
}

template <typename T>
__global__ void wg_sort_kernel(const T* in, T* out) {  // sort.cc:11-32
  __shared__ T aux[1024];
  const size_t i = threadIdx.x, wg = blockDim.x;
  aux[i] = in[i];
  __syncthreads();
  for (size_t length = 1; length < wg; length <<= 1) {
    const bool direction = ((i & (length << 1)) != 0);
    for (size_t inc = length; inc > 0; inc >>= 1) {
      const size_t j = i ^ inc;
      const T idata = aux[i], jdata = aux[j];
      const bool smaller = (jdata < idata) || (jdata == idata && j < i);
      const bool swap = smaller ^ (j < i) ^ direction;
      __syncthreads();
      aux[i] = swap ? jdata : idata;
      __syncthreads();
    }
  }
  out[i] = aux[i];
}

static inline uint32_t cap_groups(uint64_t rows) { return rows < AMMSB_MAX_GROUPS ? (uint32_t)rows : AMMSB_MAX_GROUPS; }

extern "C" int ammsb_wg_sum_f32(ammsb_ctx* ctx, const float* in, float* out, uint32_t rows, uint32_t len, uint32_t wg,
                                void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && in && out, "null argument");
  AMMSB_CHECK_ARG(ctx, wg >= 1 && wg <= 1024, "wg out of range");
  if (rows == 0) return AMMSB_OK;
  wg_sum_kernel<float><<<cap_groups(rows), wg, 0, as_stream(stream)>>>(in, out, rows, len);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

extern "C" int ammsb_wg_sum_u32(ammsb_ctx* ctx, const uint32_t* in, uint32_t* out, uint32_t rows, uint32_t len,
                                uint32_t wg, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && in && out, "null argument");
  AMMSB_CHECK_ARG(ctx, wg >= 1 && wg <= 1024, "wg out of range");
  if (rows == 0) return AMMSB_OK;
  wg_sum_kernel<uint32_t><<<cap_groups(rows), wg, 0, as_stream(stream)>>>(in, out, rows, len);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

extern "C" int ammsb_wg_normalize_f32(ammsb_ctx* ctx, float* inout, float* sums, uint32_t rows, uint32_t len,
                                      uint32_t wg, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && inout, "null argument");
  AMMSB_CHECK_ARG(ctx, wg >= 1 && wg <= 1024, "wg out of range");
  if (rows == 0) return AMMSB_OK;
  wg_normalize_kernel<<<cap_groups(rows), wg, 0, as_stream(stream)>>>(inout, sums, rows, len);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

extern "C" int ammsb_rpm_sum_f32(ammsb_ctx* ctx, const ammsb_rpm* m, float* out, uint32_t wg, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && out, "null argument");
  AMMSB_CHECK_ARG(ctx, wg >= 1 && wg <= 1024, "wg out of range");
  int rc = check_rpm(ctx, m);
  if (rc) return rc;
  rpm_sum_kernel<<<cap_groups(m->num_rows), wg, 0, as_stream(stream)>>>(*m, out, 0);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

extern "C" int ammsb_rpm_normalize_f32(ammsb_ctx* ctx, const ammsb_rpm* m, float* sums, uint32_t wg, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && sums, "null argument");
  AMMSB_CHECK_ARG(ctx, wg >= 1 && wg <= 1024, "wg out of range");
  int rc = check_rpm(ctx, m);
  if (rc) return rc;
  rpm_sum_kernel<<<cap_groups(m->num_rows), wg, 0, as_stream(stream)>>>(*m, sums, 1);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

extern "C" int ammsb_wg_sort_u32(ammsb_ctx* ctx, const uint32_t* in, uint32_t* out, uint32_t len, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && in && out, "null argument");
  AMMSB_CHECK_ARG(ctx, len >= 1 && len <= 1024 && is_pow2(len), "len must be a power of two <= 1024");
  wg_sort_kernel<uint32_t><<<1, len, 0, as_stream(stream)>>>(in, out);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

extern "C" int ammsb_wg_sort_f32(ammsb_ctx* ctx, const float* in, float* out, uint32_t len, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && in && out, "null argument");
  AMMSB_CHECK_ARG(ctx, len >= 1 && len <= 1024 && is_pow2(len), "len must be a power of two <= 1024");
  wg_sort_kernel<float><<<1, len, 0, as_stream(stream)>>>(in, out);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

__global__ void rpm_fetch_kernel(ammsb_rpm m, uint64_t row, uint64_t col, uint32_t* out) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>(rpm_row(m, row));
  out[0] = p[col];
  out[1] = p[col + 1];
}

extern "C" int ammsb_rpm_fetch(ammsb_ctx* ctx, const ammsb_rpm* m, uint64_t row, uint64_t col, uint32_t* out,
                               void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && out, "null argument");
  int rc = check_rpm(ctx, m);
  if (rc) return rc;
  AMMSB_CHECK_ARG(ctx, row < m->num_rows && col + 1 < m->num_cols, "row/col out of range");
  rpm_fetch_kernel<<<1, 1, 0, as_stream(stream)>>>(*m, row, col, out);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}

// ---- measurement aid: the shader clock each XCD holds while whatever else is running runs (DVFS give-back) ----
// One wave per block spins on the 100 MHz wall clock for `ticks` ticks and notes shader cycles / wall ticks / XCC_ID
// of its own lifetime; blocks are dealt round-robin over the XCDs, so 64 of them cover all eight.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, unsigned long long ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_readcyclecounter();
  unsigned long long w1 = w0;
  while (w1 - w0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    w1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  w1 = __builtin_amdgcn_s_memrealtime();
  out[3 * blockIdx.x + 0] = c1 - c0;
  out[3 * blockIdx.x + 1] = w1 - w0;
  out[3 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;  // HW_REG_XCC_ID
}

extern "C" int ammsb_clock_probe(ammsb_ctx* ctx, uint64_t* out, uint32_t n_blocks, uint32_t spin_us, void* stream) {
  AMMSB_CHECK_ARG(ctx, ctx && out, "null argument");
  AMMSB_CHECK_ARG(ctx, n_blocks >= 1 && n_blocks <= 4096 && spin_us >= 1 && spin_us <= 100000, "blocks in [1, 4096], spin in [1, 100000] us");
  clock_probe_kernel<<<n_blocks, 64, 0, as_stream(stream)>>>(reinterpret_cast<unsigned long long*>(out), 100ull * spin_us);
  AMMSB_LAUNCH_CHECK(ctx);
  return AMMSB_OK;
}
