// Row K: lattice kNN (replaces get_knn_3d, reference utils/torch_utils.py:16-61).
//
// The reference materialises a (375,3,5,5,5) +-1 weight on the host and a (B,375,D,H,W) conv3d output
// per call, then topk's it.  Here one 256-thread workgroup owns an 8x32 (h,w) tile of one lattice
// plane d: it stages the kernel_size planes [d-hk, d+hk] of xyz with a hk-wide halo into LDS as three
// planar arrays (zero outside the lattice == the conv's zero padding, torch_utils.py:44), and every
// lane ranks its k^3 window candidates in registers.
//
// Exactness: distances are centre - candidate, d2 = (dx*dx + dy*dy) + dz*dz in float32, no contraction
// (this file is built with -ffp-contract=off), i.e. the value torch.sum(diff**2, dim=1) produces.
// Ranking: smaller d2 first, ties by smaller candidate code (the reference's tie order is unspecified,
// SURVEY.md F10).  The top list is a sorted register array updated by one compare-exchange pass per
// accepted candidate; everything is unrolled so nothing spills to scratch.
#include <stdlib.h>

#include <utility>

#include "pf_common.h"

namespace {

// Insert (nd, nc) into the ascending list (bd, bc), dropping the last entry.  Written as a fold over
// compile-time indices: every array access has a constant index from the front end on, so the lists
// live in registers.  (A runtime-indexed `for j` version gets its selects folded into a select of the
// INDEX before unrolling, and then every access becomes a 16-way compare/cndmask chain: 5x slower.)
// new[j] = (nd < old[j-1]) ? old[j-1] : ((nd < old[j]) ? nd : old[j]); strict '<' keeps equal distances
// in arrival (= code) order.
template <int CAP, int... Js>
__device__ __forceinline__ void insert_sorted(float (&bd)[CAP], int (&bc)[CAP], float nd, int nc,
                                              std::integer_sequence<int, Js...>) {
  const float od[CAP] = {bd[Js]...};
  const int oc[CAP] = {bc[Js]...};
  const bool lt[CAP] = {(nd < od[Js])...};
  ((bd[Js] = (Js > 0 && lt[Js > 0 ? Js - 1 : 0]) ? od[Js > 0 ? Js - 1 : 0] : (lt[Js] ? nd : od[Js])), ...);
  ((bc[Js] = (Js > 0 && lt[Js > 0 ? Js - 1 : 0]) ? oc[Js > 0 ? Js - 1 : 0] : (lt[Js] ? nc : oc[Js])), ...);
}

struct Strides5 {
  int64_t b, c, d, h, w;
};

constexpr int TW = 32;
constexpr int TH = 8;

template <int CAP>
__global__ __launch_bounds__(256) void knn_lattice_kernel(const float* __restrict__ xyz, Strides5 st, int D,
                                                          int H, int W, int ks, int knn,
                                                          int64_t* __restrict__ idx_out,
                                                          uint8_t* __restrict__ code_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int hk = ks >> 1;
  const int LW = TW + 2 * hk;
  const int LH = TH + 2 * hk;
  const int plane = LH * LW;
  const int total = ks * plane;
  float* lx = lds;
  float* ly = lds + total;
  float* lz = lds + 2 * total;

  const int tx = threadIdx.x & (TW - 1);
  const int ty = threadIdx.x >> 5;
  const int w0 = blockIdx.x * TW;
  const int h0 = blockIdx.y * TH;
  const int b = blockIdx.z / D;
  const int d = blockIdx.z - b * D;

  for (int e = threadIdx.x; e < total; e += 256) {
    const int pl = e / plane;
    const int rem = e - pl * plane;
    const int r = rem / LW;
    const int cc = rem - r * LW;
    const int dd = d - hk + pl, hh = h0 - hk + r, ww = w0 - hk + cc;
    const bool in = (dd >= 0) && (dd < D) && (hh >= 0) && (hh < H) && (ww >= 0) && (ww < W);
    float vx = 0.0f, vy = 0.0f, vz = 0.0f;
    if (in) {
      const int64_t off = b * st.b + dd * st.d + hh * st.h + ww * st.w;
      vx = xyz[off];
      vy = xyz[off + st.c];
      vz = xyz[off + 2 * st.c];
    }
    lx[e] = vx;
    ly[e] = vy;
    lz[e] = vz;
  }
  __syncthreads();

  const int h = h0 + ty, w = w0 + tx;
  if (h >= H || w >= W) return;

  const int ce = (hk * LH + ty + hk) * LW + tx + hk;
  const float cx = lx[ce], cy = ly[ce], cz = lz[ce];

  // Sorted top list in registers: distances and window codes side by side.  Candidates are visited in
  // increasing code order and a candidate enters / moves up only on a STRICTLY smaller distance, so equal
  // distances keep the smaller code first -- the (d2, code) order -- with one 32-bit compare per exchange.
  float bd[CAP];
  int bc[CAP];
#pragma unroll
  for (int j = 0; j < CAP; ++j) {
    bd[j] = __builtin_huge_valf();
    bc[j] = 0;
  }

  for (int pl = 0; pl < ks; ++pl) {
    for (int r = 0; r < ks; ++r) {
      const int rowbase = (pl * LH + ty + r) * LW + tx;
      for (int cc = 0; cc < ks; ++cc) {
        const int e = rowbase + cc;
        const float dx = cx - lx[e];
        const float dy = cy - ly[e];
        const float dz = cz - lz[e];
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < bd[CAP - 1])
          insert_sorted<CAP>(bd, bc, d2, (pl * ks + r) * ks + cc, std::make_integer_sequence<int, CAP>());
      }
    }
  }

  const int64_t HW = (int64_t)H * W;
  const int64_t DHW = HW * D;
  const int64_t n = (int64_t)d * HW + (int64_t)h * W + w;
  int64_t* op = idx_out + ((int64_t)b * DHW + n) * knn;
  uint8_t* cp = code_out ? code_out + ((int64_t)b * DHW + n) * knn : nullptr;
  const int ks2 = ks * ks;
#pragma unroll
  for (int j = 0; j < CAP; ++j) {
    if (j < knn) {
      const int code = bc[j];
      const int pd = code / ks2;
      const int rem = code - pd * ks2;
      const int ph = rem / ks;
      const int pw = rem - ph * ks;
      int64_t v = n + (int64_t)(pd - hk) * HW + (int64_t)(ph - hk) * W + (pw - hk);
      v = v < 0 ? 0 : (v > DHW - 1 ? DHW - 1 : v);
      op[j] = v;
      if (cp) cp[j] = (uint8_t)code;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Split variant (knn <= 16): the four waves of a block scan one quarter of the window codes each for the
// same 64 points (2 x 32 tile), then wave 0 merges the four sorted lists through LDS.  A PointFlow stage
// has only 25 600-102 400 points: with one lane doing all 125 candidates the first iteration is 120 blocks
// of serial work (42 us); splitting the scan gives 4x the blocks and ~1/3 of the per-lane work.  The merge
// takes the smallest head, lower quarter first on equal distance == the (d2, code) order.
// ------------------------------------------------------------------------------------------------
constexpr int SW = 32, SH = 2;

__global__ __launch_bounds__(256) void knn_lattice_split_kernel(const float* __restrict__ xyz, Strides5 st, int D,
                                                                int H, int W, int ks, int knn,
                                                                int64_t* __restrict__ idx_out,
                                                                uint8_t* __restrict__ code_out) {
  constexpr int CAP = 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int hk = ks >> 1;
  const int LW = SW + 2 * hk, LH = SH + 2 * hk;
  const int plane = LH * LW;
  const int total = ks * plane;
  float* lx = lds;
  float* ly = lds + total;
  float* lz = lds + 2 * total;
  constexpr int LS = CAP + 1;                                // odd list stride: conflict-free LDS writes
  float* ld = lds + 3 * total;                               // [4][64][17] distances
  int* lc = reinterpret_cast<int*>(ld + 4 * 64 * LS);        // [4][64][17] codes

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tx = lane & (SW - 1), ty = lane >> 5;
  const int w0 = blockIdx.x * SW, h0 = blockIdx.y * SH;
  const int b = blockIdx.z / D;
  const int d = blockIdx.z - b * D;

  for (int e = threadIdx.x; e < total; e += 256) {
    const int pl = e / plane;
    const int rem = e - pl * plane;
    const int r = rem / LW;
    const int cc = rem - r * LW;
    const int dd = d - hk + pl, hh = h0 - hk + r, ww = w0 - hk + cc;
    const bool in = (dd >= 0) && (dd < D) && (hh >= 0) && (hh < H) && (ww >= 0) && (ww < W);
    float vx = 0.0f, vy = 0.0f, vz = 0.0f;
    if (in) {
      const int64_t off = b * st.b + dd * st.d + hh * st.h + ww * st.w;
      vx = xyz[off];
      vy = xyz[off + st.c];
      vz = xyz[off + 2 * st.c];
    }
    lx[e] = vx;
    ly[e] = vy;
    lz[e] = vz;
  }
  __syncthreads();

  const int h = h0 + ty, w = w0 + tx;
  const bool valid = (h < H) && (w < W);
  const int ce = (hk * LH + ty + hk) * LW + tx + hk;
  const float cx = lx[ce], cy = ly[ce], cz = lz[ce];

  float bd[CAP];
  int bc[CAP];
#pragma unroll
  for (int j = 0; j < CAP; ++j) {
    bd[j] = __builtin_huge_valf();
    bc[j] = 0x7fffffff;
  }
  const int k3 = ks * ks * ks;
  const int per = (k3 + 3) >> 2;
  const int c0 = wave * per, c1 = min(k3, c0 + per);
  const int ks2 = ks * ks;
  for (int code = c0; code < c1; ++code) {
    const int pl = code / ks2;
    const int rem = code - pl * ks2;
    const int r = rem / ks, cc = rem - r * ks;
    const int e = (pl * LH + ty + r) * LW + tx + cc;
    const float dx = cx - lx[e];
    const float dy = cy - ly[e];
    const float dz = cz - lz[e];
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (d2 < bd[CAP - 1]) insert_sorted<CAP>(bd, bc, d2, code, std::make_integer_sequence<int, CAP>());
  }
#pragma unroll
  for (int j = 0; j < CAP; ++j) {
    ld[(wave * 64 + lane) * LS + j] = bd[j];
    lc[(wave * 64 + lane) * LS + j] = bc[j];
  }
  __syncthreads();
  if (wave != 0 || !valid) return;

  // 4-way merge of the sorted lists (entries past a list's real length are +inf)
  int pos[4] = {0, 0, 0, 0};
  float hd[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) hd[q] = ld[(q * 64 + lane) * LS];
  const int64_t HW = (int64_t)H * W;
  const int64_t DHW = HW * D;
  const int64_t n = (int64_t)d * HW + (int64_t)h * W + w;
  int64_t* op = idx_out + ((int64_t)b * DHW + n) * knn;
  uint8_t* cp = code_out ? code_out + ((int64_t)b * DHW + n) * knn : nullptr;
  for (int j = 0; j < knn; ++j) {
    int best = 0;
    float bv = hd[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      if (hd[q] < bv) {            // strict: on equal distance the lower quarter (smaller codes) wins
        bv = hd[q];
        best = q;
      }
    }
    const int slot = (best * 64 + lane) * LS + pos[best];
    const int code = lc[slot];
    pos[best] += 1;
    hd[best] = pos[best] < CAP ? ld[slot + 1] : __builtin_huge_valf();
    const int pd = code / ks2;
    const int rem = code - pd * ks2;
    const int ph = rem / ks;
    const int pw = rem - ph * ks;
    int64_t v = n + (int64_t)(pd - hk) * HW + (int64_t)(ph - hk) * W + (pw - hk);
    v = v < 0 ? 0 : (v > DHW - 1 ? DHW - 1 : v);
    op[j] = v;
    if (cp) cp[j] = (uint8_t)code;
  }
}

// ------------------------------------------------------------------------------------------------
// Sorting-network variant (window 3 or 5, knn <= 16): the path the PointFlow stage uses.
//
// The insertion lists above are data dependent: a wave executes the 16-deep insert whenever ANY of its 64
// lanes accepts a candidate, i.e. for nearly all 125 candidates (68.7 us on the 102 400-point lattice of
// BASELINE config 2, 2.6 % of the HBM roof for 14 MB of traffic).  Here a lane ranks its candidates with
// branch-free compare-exchange networks on 64-bit keys (float bits of d2 << 32 | window code): d2 >= 0, so
// the unsigned order of the bits is the order of the floats, and the code in the low word resolves equal
// distances towards the smaller code -- exactly the (d2, code) order of the kernels above and of the NumPy
// brute force.  The keys are held as DOUBLES with those bits: for sign-bit-clear, non-NaN patterns (the
// float exponent lands in the double's exponent field below 0x7ff; d2 = 0 gives a denormal, and f64
// denormals are not flushed) the double order IS the unsigned order, and a compare-exchange is then
// v_min_f64 + v_max_f64 -- two instructions instead of a 64-bit compare and four conditional moves.  Candidates are taken 16 at a time (codes 16g .. 16g+15, every LDS address an immediate):
// a bitonic sort of the 16 (80 exchanges), then a truncated bitonic merge with the running best 16
// (min of list i and reversed list 15-i gives the 16 smallest as a bitonic sequence; 4 more stages sort it).
// ~1000 exchanges of 5 VALU instructions per point, no divergence, nothing in scratch.
// Outputs: the window codes as 16 bytes per point (what the EdgeConv gather passes consume) and/or the
// int64 indices of the reference API.
// ------------------------------------------------------------------------------------------------
typedef double key64;

__device__ __forceinline__ key64 make_key(float d2, int code) {
  return __longlong_as_double(((long long)__float_as_uint(d2) << 32) | (long long)code);
}
__device__ __forceinline__ unsigned key_code(key64 k) { return (unsigned)__double_as_longlong(k) & 255u; }

// plain v_min_f64 / v_max_f64 (no canonicalisation of the inputs: they are never NaN)
__device__ __forceinline__ key64 kmin(key64 a, key64 b) {
  key64 r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ key64 kmax(key64 a, key64 b) {
  key64 r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void cmpex(key64& a, key64& b) {   // afterwards a <= b
  const key64 lo = kmin(a, b), hi = kmax(a, b);
  a = lo;
  b = hi;
}

// ascending bitonic sort of 16 keys (all indices compile-time after unrolling)
__device__ __forceinline__ void sort16(key64 (&v)[16]) {
#pragma unroll
  for (int k = 2; k <= 16; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int l = i ^ j;
        if (l > i) {
          if ((i & k) == 0) cmpex(v[i], v[l]);
          else cmpex(v[l], v[i]);
        }
      }
    }
  }
}

// best <- the 16 smallest of (best U v), ascending; both inputs ascending
__device__ __forceinline__ void merge16(key64 (&best)[16], const key64 (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    best[i] = kmin(best[i], v[15 - i]);
  }
#pragma unroll
  for (int j = 8; j > 0; j >>= 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int l = i ^ j;
      if (l > i) cmpex(best[i], best[l]);
    }
  }
}

template <int KS, int ROWS>   // block = ROWS x 32 points of one lattice plane, 32*ROWS threads
__global__ __launch_bounds__(32 * ROWS) void knn_net_kernel(const float* __restrict__ xyz, Strides5 st, int D, int H,
                                                            int W, int knn, int64_t* __restrict__ idx_out,
                                                            uint8_t* __restrict__ code_out) {
  constexpr int HK = KS / 2;
  constexpr int LW = 32 + 2 * HK, LH = ROWS + 2 * HK;
  constexpr int PLANE = LH * LW, TOTAL = KS * PLANE;
  constexpr int K3 = KS * KS * KS;
  constexpr int GROUPS = (K3 + 15) / 16;
  __shared__ __attribute__((aligned(16))) float lds3[3 * TOTAL];      // one array: the index staging below reuses all of it
  float* lx = lds3;
  float* ly = lds3 + TOTAL;
  float* lz = lds3 + 2 * TOTAL;

  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int w0 = blockIdx.x * 32, h0 = blockIdx.y * ROWS;
  const int b = blockIdx.z / D;
  const int d = blockIdx.z - b * D;

  for (int e = threadIdx.x; e < TOTAL; e += 32 * ROWS) {
    const int pl = e / PLANE;
    const int rem = e - pl * PLANE;
    const int r = rem / LW;
    const int cc = rem - r * LW;
    const int dd = d - HK + pl, hh = h0 - HK + r, ww = w0 - HK + cc;
    const bool in = (dd >= 0) && (dd < D) && (hh >= 0) && (hh < H) && (ww >= 0) && (ww < W);
    float vx = 0.0f, vy = 0.0f, vz = 0.0f;
    if (in) {
      const int64_t off = b * st.b + dd * st.d + hh * st.h + ww * st.w;
      vx = xyz[off];
      vy = xyz[off + st.c];
      vz = xyz[off + 2 * st.c];
    }
    lx[e] = vx;
    ly[e] = vy;
    lz[e] = vz;
  }
  __syncthreads();

  const int h = h0 + ty, w = w0 + tx;
  const bool live = h < H && w < W;                       // (dead lanes compute on their window too: the block syncs below)
  const int base = ty * LW + tx;                          // window origin of this lane
  const int ce = base + (HK * LH + HK) * LW + HK;
  const float cx = lx[ce], cy = ly[ce], cz = lz[ce];

  key64 best[16];
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {
    key64 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int code = 16 * g + i;
      if (code < K3) {
        const int pl = code / (KS * KS), r = (code / KS) % KS, cc = code % KS;
        const int e = base + (pl * LH + r) * LW + cc;
        const float dx = cx - lx[e];
        const float dy = cy - ly[e];
        const float dz = cz - lz[e];
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        v[i] = make_key(d2, code);
      } else {
        v[i] = make_key(__builtin_huge_valf(), 255);     // past the window: never among the 16 smallest
      }
    }
    sort16(v);
    if (g == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) best[i] = v[i];
    } else {
      merge16(best, v);
    }
  }

  const int64_t HW = (int64_t)H * W;
  const int64_t DHW = HW * D;
  const int64_t n = (int64_t)d * HW + (int64_t)h * W + w;
  if (code_out != nullptr && live) {
    uint8_t* cp = code_out + ((int64_t)b * DHW + n) * knn;
    if (knn == 16) {
      unsigned pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        pk[q] = key_code(best[4 * q]) | (key_code(best[4 * q + 1]) << 8) | (key_code(best[4 * q + 2]) << 16) |
                (key_code(best[4 * q + 3]) << 24);
      *reinterpret_cast<uint4*>(cp) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j < knn) cp[j] = (uint8_t)key_code(best[j]);
    }
  }
  if (idx_out == nullptr) return;                         // (block-uniform)
  if (knn == 16) {
    // A lane's 16 indices are 128 contiguous bytes and its neighbour lane's start 128 bytes on: stored per lane, every
    // store instruction touched 32-64 cache lines (65 us for 25 600 points where the code path takes 15).  Staged through
    // LDS in chunks of J neighbours, a row of 32 points leaves as contiguous 16-byte pieces.
    constexpr int JMAX = (3 * TOTAL * 4) / (ROWS * 32 * 8);
    constexpr int J = JMAX >= 16 ? 16 : (JMAX >= 8 ? 8 : (JMAX >= 4 ? 4 : 2));
    static_assert(JMAX >= 2, "index staging does not fit the window's LDS");
    int64_t* ob = reinterpret_cast<int64_t*>(lds3);
    const int64_t row0 = (int64_t)b * DHW + (int64_t)d * HW + (int64_t)h * W + w0;      // first point of this lane's row
#pragma unroll
    for (int j0 = 0; j0 < 16; j0 += J) {
      __syncthreads();                                    // the window (or the previous chunk) has been read
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        const int code = (int)key_code(best[j0 + jj]);
        const int pd = code / (KS * KS), ph = (code / KS) % KS, pw = code % KS;
        int64_t v = n + (int64_t)(pd - HK) * HW + (int64_t)(ph - HK) * W + (pw - HK);
        v = v < 0 ? 0 : (v > DHW - 1 ? DHW - 1 : v);
        ob[(ty * 32 + tx) * J + jj] = v;
      }
      __syncthreads();
      if (h < H) {
#pragma unroll
        for (int i = 0; i < J / 2; ++i) {
          const int piece = tx + 32 * i;                  // 16-byte pieces of this row's 32 x J block
          const int pt = piece / (J / 2), q = piece - pt * (J / 2);
          if (w0 + pt < W) {
            const longlong2 v2 = *reinterpret_cast<const longlong2*>(ob + (ty * 32 + pt) * J + 2 * q);
            *reinterpret_cast<longlong2*>(idx_out + (row0 + pt) * 16 + j0 + 2 * q) = v2;
          }
        }
      }
    }
    return;
  }
  if (live) {
    int64_t* op = idx_out + ((int64_t)b * DHW + n) * knn;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < knn) {
        const int code = (int)key_code(best[j]);
        const int pd = code / (KS * KS), ph = (code / KS) % KS, pw = code % KS;
        int64_t v = n + (int64_t)(pd - HK) * HW + (int64_t)(ph - HK) * W + (pw - HK);
        v = v < 0 ? 0 : (v > DHW - 1 ? DHW - 1 : v);
        op[j] = v;
      }
    }
  }
}

}  // namespace

extern "C" int pf_knn_lattice_f32(const float* xyz, const int64_t* strides_host, int64_t B, int64_t D, int64_t H,
                                  int64_t W, int kernel_size, int knn, int64_t* idx_out, uint8_t* code_out,
                                  void* stream) {
  PF_REQUIRE(B >= 0 && D >= 0 && H >= 0 && W >= 0);
  PF_REQUIRE(kernel_size >= 1 && (kernel_size % 2) == 1);  // reference asserts odd (torch_utils.py:24)
  if (kernel_size > 7) return PF_ERR_UNSUPPORTED;  // LDS window <= 64 KiB
  const int k3 = kernel_size * kernel_size * kernel_size;
  PF_REQUIRE(knn >= 1 && knn <= k3);                       // topk would raise for knn > k^3
  if (knn > 32) return PF_ERR_UNSUPPORTED;
  if (code_out != nullptr && k3 > 256) return PF_ERR_UNSUPPORTED;
  if (B == 0 || D == 0 || H == 0 || W == 0) return PF_OK;
  PF_REQUIRE(xyz != nullptr && strides_host != nullptr && (idx_out != nullptr || code_out != nullptr));
  PF_REQUIRE(B * D <= 65535 && pf_cdiv(H, TH) <= 65535);
  PF_REQUIRE(D * H * W <= (int64_t)INT32_MAX * 16);
  Strides5 st{strides_host[0], strides_host[1], strides_host[2], strides_host[3], strides_host[4]};
  const int hk = kernel_size / 2;
  const size_t lds_bytes = (size_t)3 * kernel_size * (TH + 2 * hk) * (TW + 2 * hk) * sizeof(float);
  dim3 grid((unsigned)pf_cdiv(W, TW), (unsigned)pf_cdiv(H, TH), (unsigned)(B * D));
  hipStream_t s = (hipStream_t)stream;
  if (knn <= 16 && (kernel_size == 3 || kernel_size == 5)) {
    // sorting-network kernel; 64-point blocks while the lattice is too small to fill the chip with 256-point ones
    const bool small = B * D * H * W < 65536;
    const int rows = small ? 2 : 8;
    PF_REQUIRE(pf_cdiv(H, rows) <= 65535);
    dim3 gridn((unsigned)pf_cdiv(W, 32), (unsigned)pf_cdiv(H, rows), (unsigned)(B * D));
#define PF_KNN_NET(KSV, RV)                                                                                  \
  hipLaunchKernelGGL((knn_net_kernel<KSV, RV>), gridn, dim3(32 * RV), 0, s, xyz, st, (int)D, (int)H, (int)W, knn, \
                     idx_out, code_out)
    if (kernel_size == 5) {
      if (small) PF_KNN_NET(5, 2); else PF_KNN_NET(5, 8);
    } else {
      if (small) PF_KNN_NET(3, 2); else PF_KNN_NET(3, 8);
    }
#undef PF_KNN_NET
    return pf_launch_status();
  }
  PF_REQUIRE(idx_out != nullptr);          // the insertion-list kernels below always write the indices
  // Measured (profiles/archive/r01/r01n_microbench_knn.log, window 5, k 16): the split scan wins while the lattice is too
  // small to fill the chip with one lane per point (25 600 points: 27 vs 35 us); on 102 400 points the plain
  // scan does the same work in a quarter of the waves without the merge (49 vs 82 us).
  if (knn <= 16 && k3 >= 64 && B * D * H * W < 65536) {
    // split scan (see knn_lattice_split_kernel); every quarter holds >= 16 candidates, so the merged heads
    // never run dry before knn picks
    const size_t lds2 = (size_t)3 * kernel_size * (SH + 2 * hk) * (SW + 2 * hk) * sizeof(float) +
                        (size_t)4 * 64 * 17 * (sizeof(float) + sizeof(int));
    dim3 grid2((unsigned)pf_cdiv(W, SW), (unsigned)pf_cdiv(H, SH), (unsigned)(B * D));
    PF_REQUIRE(pf_cdiv(H, SH) <= 65535);
    hipLaunchKernelGGL(knn_lattice_split_kernel, grid2, dim3(256), lds2, s, xyz, st, (int)D, (int)H, (int)W,
                       kernel_size, knn, idx_out, code_out);
  } else if (knn <= 8) {
    hipLaunchKernelGGL(knn_lattice_kernel<8>, grid, dim3(256), lds_bytes, s, xyz, st, (int)D, (int)H, (int)W,
                       kernel_size, knn, idx_out, code_out);
  } else if (knn <= 16) {
    hipLaunchKernelGGL(knn_lattice_kernel<16>, grid, dim3(256), lds_bytes, s, xyz, st, (int)D, (int)H, (int)W,
                       kernel_size, knn, idx_out, code_out);
  } else {
    hipLaunchKernelGGL(knn_lattice_kernel<32>, grid, dim3(256), lds_bytes, s, xyz, st, (int)D, (int)H, (int)W,
                       kernel_size, knn, idx_out, code_out);
  }
  return pf_launch_status();
}
