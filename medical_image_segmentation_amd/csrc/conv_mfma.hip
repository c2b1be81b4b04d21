// 3x3 convolution as an implicit GEMM on the gfx950 matrix cores (forward and data gradient).
//
//   D[cout][pixel] += W[cout][k] * X[k][pixel],   k = (tap, cin)
//
// * Channels-last activations.  A workgroup (4 waves) owns a 16x16 pixel tile and a slice of
//   16*NT output channels, and walks tiles persistently (grid-stride), so per-channel
//   statistics are reduced once per workgroup and, when Cin fits one chunk, the weight
//   slice is staged into LDS once per workgroup instead of once per tile.
// * The 18x18xKC input halo of a tile is staged through registers into LDS (zero fill =
//   the conv's padding); every halo element is then reused by the 9 taps from LDS instead
//   of being fetched 9 times.  Global loads of stage i+1 are issued before the MFMAs of
//   stage i and written to LDS after them (loads in flight under the matrix work).
// * K is ordered (tap, cin) in 16-byte groups (8 bf16 / 4 f32 channels).  One MFMA step
//   takes 4 groups: with KC = 4 groups per position a step is one tap; with fewer channels
//   (Cin = 8, 16, 24 in bf16) a step spans several taps, so short-K layers do not pad K per
//   tap.  Weights are pre-packed in exactly the per-lane fragment order (1 KiB per
//   16-cout x 4-group fragment), so staging is a linear copy and fragment reads are
//   lane-linear ds_read_b128 (conflict-free).
// * The weight fragment is the MFMA A operand and the activation fragment the B operand, so
//   a lane's 4 accumulator registers are 4 consecutive output channels of ONE pixel: the
//   store is an 8-byte (bf16) or 16-byte (f32) vector per lane, channels-last.
// * float32 mode uses v_mfma_f32_16x16x4_f32 (exact f32 FMA chain) with the same data flow:
//   a lane's 16-byte fragment holds 4 channels and feeds 4 MFMAs.
//
// Roofline: min(MFMA, HBM) per layer.  Algorithmic bytes = B*H*W*(Cin+Cout)*sizeof(T);
// algorithmic flops = 2*9*B*H*W*Cin*Cout.
#include "conv_internal.h"
#include "reduce.h"
#include <stdlib.h>

#ifndef MIS_DBG
#define MIS_DBG 0
#endif
// -DMIS_TIMELINE=1 (tuning builds only): wave 0 of a few workgroups of the forward kernel writes
// a cycle-counter stamp at every phase boundary; mis_debug_timeline() copies the stamps out.
#ifndef MIS_TIMELINE
#define MIS_TIMELINE 0
#endif

namespace {

#if MIS_TIMELINE
__device__ unsigned long long mis_tl_buf[4][1024];
#define TL_STAMP(ID)                                                                                     \
  {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if (tl_slot >= 0 && tl_n < 1022)                                                                     \
      mis_tl_buf[tl_slot][tl_n++] = ((unsigned long long)(ID) << 56) | ((unsigned long long)clock64() & 0x00FFFFFFFFFFFFFFull); \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  }
#define TL_WAIT_LOADS() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TL_STAMP(16) }
#else
#define TL_STAMP(ID)
#define TL_WAIT_LOADS()
#endif

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// Output tile: 4 waves x MT 16-pixel row segments each, arranged as MX segments side by side
// and MT/MX rows per wave: (16/MX) rows x (16*MX) columns.  Wider tiles read longer contiguous
// runs of a channels-last row (fewer, larger DRAM bursts) for a slightly larger halo.
#ifndef MIS_CSE_14
#define MIS_CSE_14 1
#endif
#ifndef MIS_DEEP
#define MIS_DEEP 1
#endif
#ifndef MIS_PIN_FRAGS
#define MIS_PIN_FRAGS 0
#endif
#ifndef MIS_WS_FRAG2
#define MIS_WS_FRAG2 1
#endif
#ifndef MIS_AFF_REGS
#define MIS_AFF_REGS 1
#endif
#ifndef MIS_PRE_REL
#define MIS_PRE_REL 0
#endif
#ifndef MIS_FWD_MX
#define MIS_FWD_MX 1
#endif
constexpr int fwd_mx(int nt, int cpg) { return MIS_FWD_MX; }
constexpr int CM_THREADS = 256;
// rows per wave: taller tiles where the register budget allows (narrow Cout, short K chunks)
constexpr int fwd_mt(int nt, int cpg) { return 4; }   // (8-row variants measured slower: occupancy loss)

template <typename T> struct Mf;
template <> struct Mf<bf16> {
  static constexpr int G = 8;
  using Frag = bf16x8;
};
template <> struct Mf<float> {
  static constexpr int G = 4;
  using Frag = f32x4;
};

__device__ __forceinline__ void mfma_step(const bf16x8& a, const bf16x8& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mfma_step(const f32x4& a, const f32x4& b, f32x4& c) {
#pragma unroll
  for (int j = 0; j < 4; j++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
}

constexpr int nsteps_of(int cpg) { return (9 * cpg + 3) / 4; }

// relu(x*scale + shift) on a raw 16-byte channel vector (8 bf16 / 4 f32); sc/sh point at the
// vector's first channel (LDS).  Used while staging a lazily stored activation.
template <typename T>
__device__ __forceinline__ u32x4 affine_relu_raw(u32x4 raw, const float* sc, const float* sh) {
  constexpr int G = 16 / (int)sizeof(T);
  union { u32x4 u; T t[G]; } v;
  v.u = raw;
#pragma unroll
  for (int k = 0; k < G; k++) v.t[k] = from_f32<T>(fmaxf(to_f32(v.t[k]) * sc[k] + sh[k], 0.f));
  return v.u;
}

// The same with the vector's scale and shift already in registers.
template <typename T, int G>
__device__ __forceinline__ u32x4 affine_relu_regs(u32x4 raw, const float (&sc)[G], const float (&sh)[G]) {
  union { u32x4 u; T t[G]; } v;
  v.u = raw;
#pragma unroll
  for (int k = 0; k < G; k++) v.t[k] = from_f32<T>(fmaxf(to_f32(v.t[k]) * sc[k] + sh[k], 0.f));
  return v.u;
}

// Byte address of (position p, channel group cg) inside the halo image.  A position's row is
// padded so that one ds_read_b128 is conflict-free while every fragment address stays
// lane_base + a compile-time offset.  The hardware serves a b128 read in four groups of 16
// lanes: lanes {0-3, 12-15} of one 16-lane row together with lanes {4-11} of the next row.  With
// 4 groups per position (64 bytes) a 96-byte pitch puts the 8 positions of a row's part on the
// 8 even 16-byte slots of the 256-byte bank window and the other row's (channel group + 1) on
// the 8 odd ones; 1..3 groups use an odd multiple of 16 bytes (16/48/48).
template <int CPG>
constexpr int halo_pstride() { return CPG == 4 ? 96 : (CPG | 1) * 16; }
template <int CPG>
__device__ __forceinline__ int halo_addr(int p, int cg) { return p * halo_pstride<CPG>() + cg * 16; }

// Packed weight layout: [chunk][step][n-tile (16 couts)][lane 0..63][G elements]; lane l holds
// W[cout = 16n + (l&15)][k-group 4*step + (l>>4)], group -> (tap, cg) = (g / CPG, g % CPG).
// One launch packs one image, or (total2 > 0) both: elements [0, total) of the `transposed`
// image into out, then elements of the other orientation into out2.
// element i of a packed image: [chunk][step][n-tile][lane][G]
template <typename T>
__device__ __forceinline__ T pack_elem(const float* __restrict__ w, int Cw_out, int Cw_in, int tr, int cpg, int ntt,
                                       int64_t i64) {
  constexpr int G = Mf<T>::G;
  const int NS = (9 * cpg + 3) / 4;
  const int Cout = tr ? Cw_in : Cw_out;     // channels the conv produces
  const unsigned i = (unsigned)i64;         // an image has < 2^31 elements (checked where it is sized)
  const int j = (int)(i % G);
  const int l = (int)((i / G) % 64);
  const unsigned q = i / (G * 64);          // (chunk, step, n-tile)
  const int n = (int)(q % (unsigned)ntt);
  const unsigned q2 = q / (unsigned)ntt;
  const int s = (int)(q2 % (unsigned)NS);
  const int c = (int)(q2 / (unsigned)NS);
  const int g = 4 * s + (l >> 4);
  const int tap = g / cpg, cg = g % cpg;
  const int co = 16 * n + (l & 15);
  const int ci = (c * cpg + cg) * G + j;
  float v = 0.f;
  if (tap < 9 && co < Cout) {
    if (!tr) v = w[((int64_t)co * Cw_in + ci) * 9 + tap];
    else v = w[((int64_t)ci * Cw_in + co) * 9 + (8 - tap)];
  }
  return from_f32<T>(v);
}

template <typename T>
__global__ void pack_w3x3_frag_kernel(const float* __restrict__ w, int Cw_out, int Cw_in, int transposed,
                                      int CPG, int n_chunks, int nt_total, T* __restrict__ out, int64_t total,
                                      int CPG2, int n_chunks2, int nt_total2, T* __restrict__ out2, int64_t total2) {
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total + total2;
       i0 += (int64_t)gridDim.x * blockDim.x) {
    const bool second = i0 >= total;
    const int64_t i = second ? i0 - total : i0;
    const int tr = second ? !transposed : transposed;
    (second ? out2 : out)[i] = pack_elem<T>(w, Cw_out, Cw_in, tr, second ? CPG2 : CPG, second ? nt_total2 : nt_total, i);
  }
}

// Both images of MANY weight tensors in one launch: job j owns workgroups
// [first_block, first_block + n_blocks).
template <typename T>
__global__ __launch_bounds__(256) void pack_w3x3_multi_kernel(const mis_pack_job* __restrict__ jobs, int n_jobs) {
  int j = 0;
  while (j + 1 < n_jobs && (int)blockIdx.x >= jobs[j + 1].first_block) j++;
  const mis_pack_job jb = jobs[j];
  // one thread = one 16-byte fragment vector (G consecutive input channels of one (cout, tap)): the
  // index arithmetic is paid once per vector, the G sources are a strided gather, the store is 16 bytes
  constexpr int G = Mf<T>::G;
  const unsigned vf = (unsigned)(jb.total_fwd / G), vt = (unsigned)(jb.total_t / G);
  for (unsigned v0 = (unsigned)((int)blockIdx.x - jb.first_block) * 256u + threadIdx.x; v0 < vf + vt;
       v0 += (unsigned)jb.n_blocks * 256u) {
    const bool tr = v0 >= vf;
    const unsigned v = tr ? v0 - vf : v0;
    const int cpg = tr ? jb.cpg_t : jb.cpg_fwd, ntt = tr ? jb.nt_t : jb.nt_fwd;
    const int NS = (9 * cpg + 3) / 4;
    const int Cout = tr ? jb.Cw_in : jb.Cw_out;
    const int l = (int)(v % 64u);
    const unsigned q = v / 64u;
    const int n = (int)(q % (unsigned)ntt);
    const unsigned q2 = q / (unsigned)ntt;
    const int s = (int)(q2 % (unsigned)NS), c = (int)(q2 / (unsigned)NS);
    const int g = 4 * s + (l >> 4);
    const int tap = g / cpg, cg = g % cpg;
    const int co = 16 * n + (l & 15), ci0 = (c * cpg + cg) * G;
    union { u32x4 u; T t[G]; } o;
    if (tap < 9 && co < Cout) {
      const float* src = tr ? jb.w + ((int64_t)ci0 * jb.Cw_in + co) * 9 + (8 - tap)
                            : jb.w + ((int64_t)co * jb.Cw_in + ci0) * 9 + tap;
      const int64_t stride = tr ? (int64_t)jb.Cw_in * 9 : 9;
#pragma unroll
      for (int j = 0; j < G; j++) o.t[j] = from_f32<T>(src[j * stride]);
    } else {
      o.u = u32x4{0u, 0u, 0u, 0u};
    }
    *reinterpret_cast<u32x4*>((T*)(tr ? jb.out_t : jb.out_fwd) + (int64_t)v * G) = o.u;
  }
}

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// A workgroup walks a CONTIGUOUS range of tiles (x fastest, then y, then image): neighbouring
// tiles' halos overlap in the same CU's cache, and the tile coordinates advance by carries
// instead of integer divisions.
struct TilePos {
  int tx, ty, b;
  __device__ __forceinline__ void next(int tiles_x, int tiles_y) {
    if (++tx == tiles_x) {
      tx = 0;
      if (++ty == tiles_y) {
        ty = 0;
        ++b;
      }
    }
  }
};

// Everything a lane needs per stage that does not depend on the tile is computed once, so the
// per-tile cost of staging is ~4 vector instructions per 16-byte halo element, and the epilogue
// stores through buffer descriptors with precomputed lane offsets (out-of-range = dropped, so
// partial tiles and padded channels need no branches).
// workgroups per CU the register budget is held to: the light variants are latency/HBM-bound
// and want more waves in flight, the wide ones need the registers
constexpr int fwd_min_blocks(int nt, int cpg) { return (nt == 4 || nt * cpg >= 6) ? 2 : (nt == 1 && cpg <= 2 ? 4 : 3); }

// DIL: dilation of the 3x3 taps (padding = DIL, so the output keeps the input's size): the halo
// grows to (TH + 2*DIL) x (TW + 2*DIL) and the tap offsets are multiplied by DIL; nothing else changes.
//
// WS (wave-specialised, 32-channel chunks only): the workgroup has 8 waves.  Waves 4-7 are
// producers: they stage the halo and the weights of stage s+1 (global -> registers -> affine ->
// LDS) while waves 0-3, the consumers, run the MFMAs of stage s from the other of two LDS
// buffers; one barrier per stage.  A stage's vector work and its matrix work then overlap on
// every SIMD (one producer and one consumer wave each) instead of alternating: measured on the
// deep layers, the phases of the single-role kernel add up (two workgroups per CU barely
// overlap), which held them to a quarter of the MFMA rate.
template <typename T, int NT, int CPG, bool BNS, int DIL = 1, bool WS = false>
__global__ __launch_bounds__(WS ? 2 * CM_THREADS : CM_THREADS, WS ? 1 : fwd_min_blocks(NT, CPG)) void conv3x3_mfma_kernel(
    const T* __restrict__ xa, const T* __restrict__ xb, const T* __restrict__ wp, const float* __restrict__ bias,
    T* __restrict__ za, T* __restrict__ zb, float* __restrict__ partials, int B, int H, int W, int Cin, int Cs,
    int n_chunks_a, int Cout, int Cout_a, int n_chunks, int nt_total, int tiles_x, int tiles_y, int n_tiles,
    const float* __restrict__ affa, const float* __restrict__ affb, const T* __restrict__ bnz,
    const float* __restrict__ bn_aff) {
  // Input channels [0, Cs) come from xa and — two-tensor input — [Cs, Cin) from xb (same
  // channel count, chunks n_chunks_a.. of the K loop); both tensors have pixel stride Cs.
  // Output channels [0, Cout_a) go to za and — two-tensor output — [Cout_a, Cout) to zb; a
  // workgroup's channel slice lies in one of them (the host picks NT accordingly).
  constexpr int G = Mf<T>::G;
  using Frag = typename Mf<T>::Frag;
  constexpr int NS = nsteps_of(CPG);
  // DIL == 0: only the centre tap of the 3x3 weights is non-zero (a 1x1 conv carried by this kernel): no halo,
  // and of the 9 steps of a chunk (a step is a tap with 4 groups per position) only the centre one is staged
  // and multiplied
  constexpr bool CENTER = (DIL == 0);
  static_assert(!CENTER || CPG == 4, "the centre-tap form needs a step to be exactly one tap");
  constexpr int NSW = CENTER ? 1 : NS;     // steps staged and multiplied per chunk ...
  constexpr int S0 = CENTER ? 4 : 0;       // ... starting at this one of the packed image
  constexpr int KC = CPG * G;
  constexpr int MT = fwd_mt(NT, CPG);      // tile rows per wave
  constexpr int MX = fwd_mx(NT, CPG), MY = MT / MX;      // a wave's segments: MY rows x MX side by side
  constexpr int TW = 16 * MX, TH = 4 * MY;               // tile width, height
  constexpr int HW_ = TW + 2 * DIL;                      // halo width
  constexpr int HP = (TH + 2 * DIL) * HW_;               // halo positions
  constexpr int HALO_VECS = HP * CPG;
  constexpr int HALO_IT = (HALO_VECS + CM_THREADS - 1) / CM_THREADS;
  constexpr int W_VECS = NSW * NT * 64;
  constexpr int W_IT = (W_VECS + CM_THREADS - 1) / CM_THREADS;
  constexpr bool PAIR = (sizeof(T) == 2 && NT >= 2);   // 16-byte stores from lane pairs (host: Cout % 8 == 0)
  constexpr int NSTORE = PAIR ? NT / 2 : NT;
  constexpr unsigned OOB = 0x7FFFFFFFu;

  static_assert(!WS || (CPG == 4 && DIL == 1), "the wave-specialised form exists for 32-channel chunks, no dilation");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HALO_BYTES = HP * halo_pstride<CPG>() + 16;     // padded rows + 1 dummy slot
  constexpr int BUF_BYTES = HALO_BYTES + W_VECS * 16;           // one stage: halo + weight fragments
  unsigned char* halo = smem;                                   // (WS: re-pointed at the stage's buffer)
  uint4* wl = reinterpret_cast<uint4*>(smem + HALO_BYTES);
  float* biasl = reinterpret_cast<float*>(smem + (WS ? 2 : 1) * BUF_BYTES);   // [NT*16], zero beyond Cout
  float* aff = biasl + NT * 16;                                 // [2][Cin] (scale, shift) of a lazy input
  float* bnt = aff + 2 * Cin;                                   // [4][NT*16] (scale, shift, mean, rstd) of bnz
  __shared__ float red[4][2][NT * 16];

  // WS: both roles number their threads 0..255, so the staging and the fragment code below are
  // the same in either form
  const bool producer = WS && threadIdx.x >= CM_THREADS;
  const int tid = WS ? (int)(threadIdx.x & (CM_THREADS - 1)) : (int)threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;
  const int n0 = blockIdx.y * NT;                 // first n-tile of this workgroup
#if MIS_TIMELINE
  int tl_n = 0;
  const int tl_slot = (tid != 0 || blockIdx.y != 0) ? -1
                      : WS ? (blockIdx.x == 0 ? (producer ? 1 : 0) : (blockIdx.x == gridDim.x - 1 && !producer) ? 2 : -1)
                      : blockIdx.x == 0 ? 0 : blockIdx.x == gridDim.x / 2 ? 1 : blockIdx.x == gridDim.x - 1 ? 2 : -1;
  TL_STAMP(1)
#endif

  // per-lane tap offsets of each MFMA step (positions inside the halo) and channel group
  // (with 4 groups per position a step is exactly one tap: offsets are compile-time constants)
  constexpr int NTAB = (CPG == 4) ? 1 : NS;
  int tap_addr[NTAB];
  if constexpr (CPG != 4) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
      int g = 4 * s + kg;
      int tap = g / CPG;
      if (tap > 8) tap = 8;                        // padded group: weights are zero, read finite data
      tap_addr[s] = halo_addr<CPG>(((tap / 3) * HW_ + (tap % 3)) * DIL, g % CPG);
    }
  }
  // LDS byte address of this lane's first fragment: wave's first row, lane's column (and group)
  const int frag_base = halo_addr<CPG>(wid * MY * HW_ + col, CPG == 4 ? kg : 0);

  // tile-invariant staging state of this lane's halo elements: kept in registers, except in
  // the widest variant (no registers to spare) which recomputes it per stage
  constexpr bool PRE = WS || !(NT == 4 && CPG == 4);
  constexpr bool PRE_REL = PRE || (MIS_PRE_REL != 0);      // the widest variant keeps only the offsets
  int tid_l = tid;        // laundered per stage when !PRE, so the recomputation is not hoisted (and spilled)
  // (unsigned arithmetic: divisions by the compile-time halo width become a multiply-high and a shift)
  auto halo_rel = [&](int i) -> unsigned {      // byte offset of element i relative to the halo origin
    const unsigned v = (unsigned)tid_l + i * CM_THREADS, pos = v / CPG, cg = v % CPG;
    return (((pos / HW_) * (unsigned)W + pos % HW_) * (unsigned)Cs + cg * G) * (unsigned)sizeof(T);
  };
  auto halo_px = [&](int i) -> int {            // its halo column (huge for tail elements: never loaded)
    const unsigned v = (unsigned)tid_l + i * CM_THREADS;
    return v < (unsigned)HALO_VECS ? (int)((v / CPG) % HW_) : (1 << 28);
  };
  auto halo_lds = [&](int i) -> int {           // its LDS slot (the dummy slot for tail elements)
    const unsigned v = (unsigned)tid_l + i * CM_THREADS;
    return v < (unsigned)HALO_VECS ? (int)((v / CPG) * halo_pstride<CPG>() + (v % CPG) * 16) : HALO_BYTES - 16;
  };
  unsigned h_rel[PRE_REL ? HALO_IT : 1];
  int h_px[PRE ? HALO_IT : 1], h_lds[PRE ? HALO_IT : 1];
  if constexpr (PRE_REL) {
#pragma unroll
    for (int i = 0; i < HALO_IT; i++) h_rel[i] = halo_rel(i);
  }
  if constexpr (PRE) {
#pragma unroll
    for (int i = 0; i < HALO_IT; i++) {
      h_px[i] = halo_px(i);
      h_lds[i] = halo_lds(i);
    }
  }
  // lane offsets of the epilogue stores inside the tile (bytes), OOB for padded channels
  const bool zsel = zb != nullptr && n0 * 16 >= Cout_a;          // this workgroup writes the second tensor
  T* const zs = zsel ? zb : za;
  const int Cz = zb == nullptr ? Cout : (zsel ? Cout - Cout_a : Cout_a);   // its channel count
  unsigned z_rel[NSTORE];
#pragma unroll
  for (int j = 0; j < NSTORE; j++) {
    int co;
    if constexpr (PAIR) co = (n0 + 2 * j) * 16 + ((kg & 1) ? 16 + 4 * (kg - 1) : 4 * kg);
    else co = (n0 + j) * 16 + 4 * kg;
    const int cl = co - (zsel ? Cout_a : 0);
    z_rel[j] = (co < Cout) ? (unsigned)((((wid * MY) * W + col) * Cz + cl) * (int)sizeof(T)) : OOB;
  }
  const unsigned z_row = (unsigned)(W * Cz * (int)sizeof(T)), z_seg = (unsigned)(16 * Cz * (int)sizeof(T));
  // lane offset of the accumulators' own 4 channels (n-tile 0) inside the tile: for the BatchNorm-backward sums
  const unsigned bz_rel = (unsigned)((((wid * MY) * W + col) * Cz + n0 * 16 + 4 * kg) * (int)sizeof(T));

  f32x4 s1[NT], s2[NT];
#pragma unroll
  for (int n = 0; n < NT; n++) s1[n] = s2[n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DEEP (weights resident, one K chunk per tile): two register sets, the halo of tile t+2 is
  // requested while tile t computes, so a load has two tile times to arrive.  Measured on the
  // write-heavy layers: reads queued behind a saturated store path otherwise stall the pipeline
  // (time = read phase + write phase instead of their maximum).
  constexpr bool AFF_REGS = (CM_THREADS % CPG == 0) && (MIS_AFF_REGS != 0);
  constexpr bool DEEP = (CPG < 4) && (MIS_DEEP != 0) && !BNS && DIL == 1 && !WS;   // (the fused-sums epilogue twice over would spill)
  u32x4 hreg[HALO_IT], hregB[DEEP ? HALO_IT : 1], wreg[W_IT];
#define HREG hreg
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, n_chunks * NS * nt_total * 1024, 0x00020000);

  // Global loads go through buffer descriptors: an out-of-range offset returns zeros in
  // hardware.  Rows above/below the image fall out of the per-image descriptor by themselves
  // (negative or too-large offsets); only the column needs a compare.  This implements the
  // conv's zero padding without a branch or a select on the loaded value, so the loads stay in
  // flight under the MFMAs.  A dead stage (past the last tile) shifts x far left: nothing loads.
  const int img_bytes = H * W * Cs * (int)sizeof(T);
#define LOAD_HALO(P, CHUNK, LIVE)                                                                        \
  {                                                                                                      \
    const int xo_ = (P).tx * TW - DIL, yo_ = (P).ty * TH - DIL;                                          \
    const int x0_ = ((LIVE) && !(MIS_DBG & 4)) ? xo_ : -(1 << 29);                                       \
    const bool sb_ = (CHUNK) >= n_chunks_a;                                                              \
    const T* xs_ = sb_ ? xb : xa;                                                                        \
    const int cc_ = sb_ ? (CHUNK) - n_chunks_a : (CHUNK);                                                \
    const unsigned base_ = (unsigned)(((yo_ * W + xo_) * Cs + cc_ * KC) * (int)sizeof(T));               \
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                                \
        (void*)(xs_ + (int64_t)(P).b * H * W * Cs), 0, img_bytes, 0x00020000);                           \
    _Pragma("unroll") for (int i = 0; i < HALO_IT; i++) {                                                \
      const bool ok = (unsigned)(x0_ + (PRE ? h_px[PRE ? i : 0] : halo_px(i))) < (unsigned)W;            \
      const unsigned rel_ = PRE_REL ? h_rel[PRE_REL ? i : 0] : halo_rel(i);                              \
      HREG[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)(ok ? base_ + rel_ : OOB), 0, 0);        \
    }                                                                                                    \
  }
  // STORE_HALO(P, CHUNK): registers -> LDS.  For a lazily stored input the BatchNorm affine and
  // ReLU of the producing layer are applied here, to in-image positions only (padding stays 0).
#define STORE_HALO(P, CHUNK)                                                                             \
  {                                                                                                      \
    const bool lazy_ = ((CHUNK) >= n_chunks_a) ? (affb != nullptr) : (affa != nullptr);                  \
    if (AFF_REGS && lazy_) {                                                                             \
      /* every element of a thread has the same channel group (the workgroup size is a multiple of */    \
      /* CPG): its scale and shift are fetched once per stage, ahead of the wait for the halo loads, */  \
      /* and the in-image test is a select, so the elements form no chain of LDS round trips */          \
      const int c_ = (CHUNK) * KC + (tid % CPG) * G;                                                     \
      float sc_[G], sh_[G];                                                                              \
      _Pragma("unroll") for (int k = 0; k < G; k += 4) {                                                 \
        const f32x4 a_ = *reinterpret_cast<const f32x4*>(aff + c_ + k);                                  \
        const f32x4 b_ = *reinterpret_cast<const f32x4*>(aff + Cin + c_ + k);                            \
        _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                  \
          sc_[k + r] = a_[r];                                                                            \
          sh_[k + r] = b_[r];                                                                            \
        }                                                                                                \
      }                                                                                                  \
      const int gy0_ = (P).ty * TH - DIL, gx0_ = (P).tx * TW - DIL;                                      \
      _Pragma("unroll") for (int i = 0; i < HALO_IT; i++) {                                              \
        const u32x4 raw_ = HREG[i];                                                                      \
        const unsigned pos = ((unsigned)tid + i * CM_THREADS) / CPG;                                     \
        const bool in_ = (unsigned)(gy0_ + (int)(pos / HW_)) < (unsigned)H &&                            \
                         (unsigned)(gx0_ + (int)(pos % HW_)) < (unsigned)W;                              \
        const u32x4 act_ = affine_relu_regs<T, G>(raw_, sc_, sh_);                                       \
        u32x4 val;                                                                                       \
        _Pragma("unroll") for (int r = 0; r < 4; r++) val[r] = in_ ? act_[r] : raw_[r];                  \
        *reinterpret_cast<u32x4*>(halo + (PRE ? h_lds[PRE ? i : 0] : halo_lds(i))) = val;                \
      }                                                                                                  \
    } else {                                                                                             \
      _Pragma("unroll") for (int i = 0; i < HALO_IT; i++) {                                              \
        u32x4 val = HREG[i];                                                                             \
        if (lazy_) {                                                                                     \
          const int v = tid + i * CM_THREADS;                                                            \
          const int pos = v / CPG, cg = v % CPG;                                                         \
          const int gy = (P).ty * TH - DIL + pos / HW_, gx = (P).tx * TW - DIL + pos % HW_;              \
          const int c = (CHUNK) * KC + cg * G;                                                           \
          if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)                                  \
            val = affine_relu_raw<T>(val, aff + c, aff + Cin + c);                                       \
        }                                                                                                \
        *reinterpret_cast<u32x4*>(halo + (PRE ? h_lds[PRE ? i : 0] : halo_lds(i))) = val;                \
      }                                                                                                  \
    }                                                                                                    \
  }
#define LOAD_W(CHUNK, LIVE)                                                                              \
  {                                                                                                      \
    const bool live_ = (LIVE);                                                                           \
    _Pragma("unroll") for (int i = 0; i < W_IT; i++) {                                                   \
      /* element v = tid + i*256 of the chunk's [step][n-tile][lane] image; with 4 n-tiles a step */     \
      /* is exactly one pass of the workgroup, so step = i and the rest is the lane's own offset */      \
      const unsigned v = (unsigned)tid_l + i * CM_THREADS;                                               \
      const unsigned s_ = (NT * 64 == CM_THREADS) ? (unsigned)i : v / (NT * 64);                         \
      const unsigned rem = (NT * 64 == CM_THREADS) ? (unsigned)tid_l : v % (NT * 64);                    \
      const unsigned off = ((((unsigned)(CHUNK) * NS + S0 + s_) * (unsigned)nt_total + (unsigned)n0) * 64 + rem) * 16; \
      wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, (live_ && v < (unsigned)W_VECS) ? (int)off : (int)OOB, 0, 0); \
    }                                                                                                    \
  }
#define STORE_W()                                                                                        \
  {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < W_IT; i++) {                                                   \
      const int v = tid + i * CM_THREADS;                                                                \
      if (v < W_VECS) *reinterpret_cast<u32x4*>(&wl[v]) = wreg[i];                                       \
    }                                                                                                    \
  }
  // EPILOGUE(): store z channels-last and accumulate the statistics.  The accumulators
  // already hold the bias (they start from it).  A lane owns 4 consecutive channels of a pixel
  // per n-tile; in bf16 with >= 2 n-tiles, lane rows (kg, kg^1) exchange halves of a tile pair
  // (one v_permlane16_swap per register) so every lane stores 16 bytes = 8 consecutive channels.
#define EPILOGUE()                                                                                       \
  {                                                                                                      \
    bool pvm[MT];                                                                                        \
    _Pragma("unroll") for (int m = 0; m < MT; m++) pvm[m] = true;                                        \
    if (!(live && (cur.ty * TH + TH <= H) && (cur.tx * TW + TW <= W))) {                                 \
      /* partial tile (or the leading load-only iteration): pixels outside contribute nothing */         \
      _Pragma("unroll") for (int m = 0; m < MT; m++) {                                                   \
        pvm[m] = live && (cur.ty * TH + wid * MY + m / MX < H) && (cur.tx * TW + (m % MX) * 16 + col < W); \
        if (!pvm[m]) {                                                                                   \
          _Pragma("unroll") for (int n = 0; n < NT; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};          \
        }                                                                                                \
      }                                                                                                  \
    }                                                                                                    \
    if (partials && !BNS) {                                                                              \
      _Pragma("unroll") for (int m = 0; m < MT; m++) {                                                   \
        _Pragma("unroll") for (int n = 0; n < NT; n++) {                                                 \
          s1[n] += acc[m][n];                                                                            \
          s2[n] += acc[m][n] * acc[m][n];                                                                \
        }                                                                                                \
      }                                                                                                  \
    }                                                                                                    \
    if constexpr (BNS) {                                                                                 \
      /* this conv is a data gradient: its output is the gradient `da` of the lazy activation */         \
      /* relu(bn(bnz)); accumulate that BatchNorm's backward sums (sum dy, sum dy*xhat) here, */          \
      /* from the values as they are stored (rounded to T), instead of a separate pass over both */      \
      const __amdgpu_buffer_rsrc_t brs_ = __builtin_amdgcn_make_buffer_rsrc(                             \
          (void*)(bnz + (int64_t)cur.b * H * W * Cz), 0, H * W * Cz * (int)sizeof(T), 0x00020000);       \
      const unsigned bb_ = (unsigned)(((cur.ty * TH * W) + cur.tx * TW) * Cz * (int)sizeof(T));          \
      _Pragma("unroll") for (int n = 0; n < NT; n++) {                                                   \
        const f32x4 sc_ = *reinterpret_cast<const f32x4*>(&bnt[0 * NT * 16 + n * 16 + 4 * kg]);          \
        const f32x4 sh_ = *reinterpret_cast<const f32x4*>(&bnt[1 * NT * 16 + n * 16 + 4 * kg]);          \
        const f32x4 mu_ = *reinterpret_cast<const f32x4*>(&bnt[2 * NT * 16 + n * 16 + 4 * kg]);          \
        const f32x4 rs_ = *reinterpret_cast<const f32x4*>(&bnt[3 * NT * 16 + n * 16 + 4 * kg]);          \
        const bool cok_ = (n0 + n) * 16 + 4 * kg < Cout;                                                 \
        _Pragma("unroll") for (int m = 0; m < MT; m++) {                                                 \
          const unsigned o_ = bb_ + (unsigned)(m / MX) * z_row + (unsigned)(m % MX) * z_seg + bz_rel +   \
                              (unsigned)(n * 16 * (int)sizeof(T));                                       \
          const int off_ = (int)((pvm[m] && cok_) ? o_ : OOB);                                           \
          float zv_[4];                                                                                  \
          if constexpr (sizeof(T) == 2) {                                                                \
            union { u32x2 u; bf16 h[4]; } q_;                                                            \
            q_.u = __builtin_amdgcn_raw_buffer_load_b64(brs_, off_, 0, 0);                               \
            _Pragma("unroll") for (int r = 0; r < 4; r++) zv_[r] = to_f32(q_.h[r]);                      \
          } else {                                                                                       \
            union { u32x4 u; float f[4]; } q_;                                                           \
            q_.u = __builtin_amdgcn_raw_buffer_load_b128(brs_, off_, 0, 0);                              \
            _Pragma("unroll") for (int r = 0; r < 4; r++) zv_[r] = q_.f[r];                              \
          }                                                                                              \
          _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                \
            /* (a select, not a product with zero: where the gradient does not count, nothing that */    \
            /* was read for it may reach the sums — 0 * NaN is NaN) */                                   \
            const bool on_ = pvm[m] && cok_ && (zv_[r] * sc_[r] + sh_[r] > 0.f);                         \
            const float d_ = on_ ? to_f32(from_f32<T>(acc[m][n][r])) : 0.f;                              \
            s1[n][r] += d_;                                                                              \
            s2[n][r] = on_ ? __builtin_fmaf(d_, (zv_[r] - mu_[r]) * rs_[r], s2[n][r]) : s2[n][r];        \
          }                                                                                              \
        }                                                                                                \
      }                                                                                                  \
    }                                                                                                    \
    const __amdgpu_buffer_rsrc_t zrs_ = __builtin_amdgcn_make_buffer_rsrc(                               \
        (void*)(zs + (int64_t)cur.b * H * W * Cz), 0, H * W * Cz * (int)sizeof(T), 0x00020000);          \
    const unsigned zb_ = (unsigned)(((cur.ty * TH * W) + cur.tx * TW) * Cz * (int)sizeof(T));            \
    _Pragma("unroll") for (int m = 0; m < MT; m++) {                                                     \
      const unsigned zm_ = zb_ + (unsigned)(m / MX) * z_row + (unsigned)(m % MX) * z_seg;                \
      _Pragma("unroll") for (int j = 0; j < NSTORE; j++) {                                               \
        const int off_ = (int)((pvm[m] && !(MIS_DBG & 1)) ? zm_ + z_rel[j] : OOB);                       \
        if constexpr (PAIR) {                                                                            \
          union { bf16 h[4]; unsigned int u[2]; } pa, pb;                                                \
          _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                \
            pa.h[r] = from_f32<bf16>(acc[m][2 * j][r]);                                                  \
            pb.h[r] = from_f32<bf16>(acc[m][2 * j + 1][r]);                                              \
          }                                                                                              \
          const u32x2 q0 = __builtin_amdgcn_permlane16_swap(pa.u[0], pb.u[0], false, false);             \
          const u32x2 q1 = __builtin_amdgcn_permlane16_swap(pa.u[1], pb.u[1], false, false);             \
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{q0[0], q1[0], q0[1], q1[1]}, zrs_, off_, 0, 0);   \
        } else if constexpr (sizeof(T) == 2) {                                                           \
          union { bf16 h[4]; u32x2 u; } pa;                                                              \
          _Pragma("unroll") for (int r = 0; r < 4; r++) pa.h[r] = from_f32<bf16>(acc[m][j][r]);          \
          __builtin_amdgcn_raw_buffer_store_b64(pa.u, zrs_, off_, 0, 0);                                 \
        } else {                                                                                         \
          union { f32x4 f; u32x4 u; } pa;                                                                \
          pa.f = acc[m][j];                                                                              \
          __builtin_amdgcn_raw_buffer_store_b128(pa.u, zrs_, off_, 0, 0);                                \
        }                                                                                                \
      }                                                                                                  \
    }                                                                                                    \
  }

  // this workgroup's contiguous tile range
  const int nb_ = gridDim.x, q_ = n_tiles / nb_, r_ = n_tiles % nb_;
  const int bx = blockIdx.x;
  const int t_begin = bx * q_ + (bx < r_ ? bx : r_);
  const int t_count = q_ + (bx < r_ ? 1 : 0);
  TilePos cur;
  cur.tx = t_begin % tiles_x;
  cur.ty = (t_begin / tiles_x) % tiles_y;
  cur.b = t_begin / (tiles_x * tiles_y);

  if constexpr (WS) {
    // the producers' first requests go out before the tables below are filled
    if (producer) {
      LOAD_HALO(cur, 0, t_count > 0)
      LOAD_W(0, t_count > 0)
    }
  }
  for (int i = tid; i < NT * 16; i += CM_THREADS) {
    const int co = n0 * 16 + i;
    biasl[i] = (bias && co < Cout) ? bias[co] : 0.f;
  }
  // combined affine table [2][Cin] of the (up to two) lazy inputs; visible after the first barrier below
  if (affa) {
    for (int i = tid; i < 2 * Cs; i += CM_THREADS) aff[(i / Cs) * Cin + i % Cs] = affa[i];
  }
  if (affb) {
    const int Cb = Cin - Cs;
    for (int i = tid; i < 2 * Cb; i += CM_THREADS) aff[(i / Cb) * Cin + Cs + i % Cb] = affb[i];
  }
  if constexpr (BNS) {
    for (int i = tid; i < 4 * NT * 16; i += CM_THREADS) {
      const int co = n0 * 16 + i % (NT * 16);
      bnt[i] = (co < Cout) ? bn_aff[(i / (NT * 16)) * Cout + co] : 0.f;
    }
  }
  if constexpr (BNS && !WS) {
    // The single-role loop starts with a load-only iteration whose epilogue already reads this table,
    // before any barrier of the loop: without this one a wave could read entries that another wave has
    // not written yet, i.e. whatever the LDS held before (harmless only while that is this kernel's
    // own earlier table; NaNs once another kernel's workgroups use the same CU in between).
    __syncthreads();
  }
  // Weights that fit one chunk stay in LDS for the whole kernel.  With fewer than 4 groups per
  // position that is always so and they are staged here, once; otherwise (runtime chunk count)
  // they travel through the stage pipeline below.
  constexpr bool W_ONCE = (CPG < 4);
  const bool w_resident = (n_chunks == 1);
  if constexpr (W_ONCE) {
    LOAD_W(0, true)
    STORE_W()                               // visible after the first barrier of the tile loop
  }
  bool first = true;

#define LOAD_FRAGS(S, BUF)                                                                               \
  {                                                                                                      \
    _Pragma("unroll") for (int m = 0; m < MT; m++) {                                                     \
      int a_;                                                                                            \
      const int mp_ = (m / MX) * HW_ + (m % MX) * 16;     /* segment m: row m/MX, columns 16*(m%MX).. */ \
      if constexpr (CPG == 4) a_ = frag_base + halo_addr<CPG>(mp_ + (((S) / 3) * HW_ + ((S) % 3)) * DIL, 0); \
      else a_ = frag_base + tap_addr[CPG == 4 ? 0 : (S)] + halo_addr<CPG>(mp_, 0);                       \
      xa[BUF][m] = *reinterpret_cast<const Frag*>(halo + a_);                                            \
    }                                                                                                    \
    _Pragma("unroll") for (int n = 0; n < NT; n++)                                                       \
      wf[BUF][n] = *reinterpret_cast<const Frag*>(&wl[((S) * NT + n) * 64 + lane]);                      \
  }
  // the MFMAs of one stage (accumulators start from the bias on a tile's first chunk)
#define STAGE_MFMA(FIRST_CHUNK)                                                                          \
    if (live) {                                                                                          \
      if (FIRST_CHUNK) {                                                                                 \
        _Pragma("unroll") for (int n = 0; n < NT; n++) {                                                 \
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(&biasl[n * 16 + 4 * kg]);                     \
          _Pragma("unroll") for (int m = 0; m < MT; m++) acc[m][n] = b4;                                 \
        }                                                                                                \
      }                                                                                                  \
      Frag xa[2][MT], wf[2][NT];                                                                         \
      if constexpr ((NT < 4 || (WS && MIS_WS_FRAG2)) && !(MIS_CSE_14 && NT == 1 && CPG == 4)) {             \
        /* fragments of step s+1 are fetched from LDS while the MFMAs of step s run (two register */     \
        /* sets); the scheduling barrier keeps the compiler from hoisting every step's reads up */       \
        LOAD_FRAGS(0, 0)                                                                                 \
        _Pragma("unroll") for (int s = 0; s < ((MIS_DBG & 8) ? 1 : NSW); s++) {                          \
          if (s + 1 < NSW) LOAD_FRAGS(s + 1, (s + 1) & 1)                                                \
          if constexpr (WS || MIS_PIN_FRAGS) __builtin_amdgcn_sched_barrier(0); /* reads first: a step of MFMAs covers their latency */ \
          _Pragma("unroll") for (int n = 0; n < NT; n++)                                                 \
            _Pragma("unroll") for (int m = 0; m < MT; m++)                                               \
              mfma_step(wf[s & 1][n], xa[s & 1][m], acc[m][n]);                                          \
          __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                \
      } else {                                                                                           \
        /* widest variants: one fragment set, the compiler interleaves reads and MFMAs itself */         \
        _Pragma("unroll") for (int s = 0; s < NSW; s++) {                                                \
          LOAD_FRAGS(s, 0)                                                                               \
          _Pragma("unroll") for (int n = 0; n < NT; n++)                                                 \
            _Pragma("unroll") for (int m = 0; m < MT; m++) mfma_step(wf[0][n], xa[0][m], acc[m][n]);     \
        }                                                                                                \
      }                                                                                                  \
    }

  // One stage = one (tile, 32-channel chunk): halo (+ weights) registers -> LDS, issue the next
  // stage's global loads, then the MFMAs.  The loads are issued unconditionally (dead stages
  // read out of range = zeros).  FIRST_CHUNK: the accumulators start from the bias.
#define STAGE(CHUNK, FIRST_CHUNK)                                                                        \
  {                                                                                                      \
    TL_STAMP(10)                                                                                         \
    if (live) {                                                                                          \
      __syncthreads(); /* every wave is done reading the previous stage */                               \
      TL_STAMP(11)                                                                                       \
      TL_WAIT_LOADS()                                                                                    \
      if constexpr (!PRE) asm volatile("" : "+v"(tid_l));                                                \
      STORE_HALO(cur, (CHUNK))                                                                           \
      if constexpr (!W_ONCE) {                                                                           \
        if (!w_resident || first) STORE_W()                                                              \
        first = false;                                                                                   \
      }                                                                                                  \
      TL_STAMP(12)                                                                                       \
      __syncthreads();                                                                                   \
      TL_STAMP(13)                                                                                       \
    }                                                                                                    \
    if constexpr (!PRE) asm volatile("" : "+v"(tid_l));                                                  \
    if ((CHUNK) + 1 < nch) {                                                                             \
      LOAD_HALO(cur, (CHUNK) + 1, true)                                                                  \
      if constexpr (!W_ONCE) LOAD_W((CHUNK) + 1, true)                                                   \
    } else {                                                                                             \
      LOAD_HALO(nxt, 0, it + 1 < t_count)                                                                \
      if constexpr (!W_ONCE) LOAD_W(0, (!w_resident || !live) && it + 1 < t_count)                       \
    }                                                                                                    \
    __builtin_amdgcn_sched_barrier(0); /* loads stay ahead of the MFMAs (the scheduler would sink them) */ \
    TL_STAMP(14)                                                                                         \
    STAGE_MFMA(FIRST_CHUNK)                                                                              \
    TL_STAMP(15)                                                                                         \
  }

  // The loop runs one extra, leading iteration (it = -1) that only issues the first tile's
  // loads and an epilogue whose stores are all out of range.  No load is in flight when the
  // loop is entered, so every path into a tile's first wait has the same shape — loads, then
  // an epilogue's stores — and the wait the compiler derives covers the loads only: a tile
  // does not wait for the previous tile's stores to be acknowledged (the counter retires in
  // order and the stores are younger than the loads).
  TL_STAMP(2)
  TilePos nxt = cur;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int n = 0; n < NT; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (WS) {
    // Stage s = (tile, chunk) number s of this workgroup's range, in LDS buffer s & 1.
    const int S = t_count * n_chunks;
#define SET_BUF(BUFI)                                                                                    \
  {                                                                                                      \
    halo = smem + (BUFI) * BUF_BYTES;                                                                    \
    wl = reinterpret_cast<uint4*>(halo + HALO_BYTES);                                                    \
  }
    __syncthreads();          // the tables above (affine, bias) are read by the first stage's staging
    if (producer) {
      // `ld`: the stage whose loads are in flight (or next to be issued); the stage being stored
      // is the one loaded last, so its position is `ld` before the advance
      TilePos ld = cur;
      int ldc = 0;
#define ADVANCE_LD()                                                                                     \
  if (++ldc == n_chunks) {                                                                               \
    ldc = 0;                                                                                             \
    ld.next(tiles_x, tiles_y);                                                                           \
  }
      SET_BUF(0)                 /* (stage 0 was requested before the tables were filled) */
      STORE_HALO(ld, ldc)
      STORE_W()
      ADVANCE_LD()
      LOAD_HALO(ld, ldc, 1 < S)
      LOAD_W(ldc, 1 < S)
      __syncthreads();
      for (int s = 0; s < S; s++) {
        TL_STAMP(10)
        if (s + 1 < S) {
          TL_WAIT_LOADS()
          SET_BUF((s + 1) & 1)
          STORE_HALO(ld, ldc)
          // weights that fit one chunk are the same in every stage: both buffers hold them after two stages
          if (!w_resident || s == 0) STORE_W()
        }
        TL_STAMP(12)
        ADVANCE_LD()
        LOAD_HALO(ld, ldc, s + 2 < S)
        LOAD_W(ldc, !w_resident && s + 2 < S)
        TL_STAMP(14)
        __syncthreads();
      }
#undef ADVANCE_LD
    } else {
      const bool live = true;
      int chunk = 0;
      __syncthreads();
      for (int s = 0; s < S; s++) {
        TL_STAMP(14)
        SET_BUF(s & 1)
        if (chunk == 0) {
          STAGE_MFMA(true)
        } else {
          STAGE_MFMA(false)
        }
        TL_STAMP(15)
        __syncthreads();       // the producers may overwrite this buffer; the epilogue below needs no LDS buffer
        TL_STAMP(13)
        if (++chunk == n_chunks) {
          chunk = 0;
          EPILOGUE()
          cur.next(tiles_x, tiles_y);
          TL_STAMP(20)
        }
      }
    }
#undef SET_BUF
  } else if constexpr (DEEP) {
    // Two tiles per trip, one per register set; two leading load-only halves (it = -2, -1) give
    // every path into a wait the same shape, as above.  `cur` = the tile computed, `nxt` = the
    // tile requested (two ahead).
#define HALF(IDX)                                                                                        \
  {                                                                                                      \
    const int idx_ = (IDX);                                                                              \
    const bool live = idx_ >= 0 && idx_ < t_count;                                                       \
    TL_STAMP(10)                                                                                         \
    if (live) {                                                                                          \
      __syncthreads();                                                                                   \
      TL_STAMP(11)                                                                                       \
      TL_WAIT_LOADS()                                                                                    \
      STORE_HALO(cur, 0)                                                                                 \
      TL_STAMP(12)                                                                                       \
      __syncthreads();                                                                                   \
      TL_STAMP(13)                                                                                       \
    }                                                                                                    \
    LOAD_HALO(nxt, 0, idx_ + 2 < t_count)                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    TL_STAMP(14)                                                                                         \
    STAGE_MFMA(true)                                                                                     \
    TL_STAMP(15)                                                                                         \
    EPILOGUE()                                                                                           \
    TL_STAMP(20)                                                                                         \
    if (live) cur.next(tiles_x, tiles_y);                                                                \
    nxt.next(tiles_x, tiles_y);                                                                          \
  }
    for (int it = -2; it < t_count; it += 2) {
      HALF(it)
#undef HREG
#define HREG hregB
      HALF(it + 1)
#undef HREG
#define HREG hreg
    }
#undef HALF
  } else {
    for (int it = -1; it < t_count; it++) {
      const bool live = it >= 0;
      const int nch = live ? n_chunks : 1;
      if (live) nxt.next(tiles_x, tiles_y);
      // the first chunk is peeled: its wait sits behind an epilogue's stores, the others' do not
      STAGE(0, true)
      for (int chunk = 1; chunk < nch; chunk++) STAGE(chunk, false)
      EPILOGUE()
      TL_STAMP(20)
      cur = nxt;
    }
  }
#undef STAGE
#undef STAGE_MFMA
#undef HREG
#undef LOAD_FRAGS

  if (partials) {
    // sum over the 16 pixel lanes that share a channel quad, then over the 4 (consumer) waves
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float a = row16_sum(s1[n][r]), q = row16_sum(s2[n][r]);
        if (col == 0 && !producer) {
          red[wid][0][n * 16 + 4 * kg + r] = a;
          red[wid][1][n * 16 + 4 * kg + r] = q;
        }
      }
    __syncthreads();
    if (tid < 2 * NT * 16 && !producer) {
      const int which = tid / (NT * 16), c = tid % (NT * 16);
      const float v = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
      const int co = n0 * 16 + c;
      if (co < Cout) partials[((int64_t)blockIdx.x * 2 + which) * Cout + co] = v;
    }
  }
#if MIS_TIMELINE
  TL_STAMP(30)
  if (tl_slot >= 0) mis_tl_buf[tl_slot][1023] = (unsigned long long)tl_n;
#endif
}

#undef LOAD_HALO
#undef STORE_HALO
#undef LOAD_W
#undef STORE_W
#undef EPILOGUE

struct FwdPlan {
  int CPG, NT, n_chunks, nt_total, grid_x, grid_y, tiles_x, tiles_y, n_tiles;
  size_t lds;
  bool ws;        // wave-specialised form (8 waves, two LDS stage buffers, one workgroup per CU)
};

// Workgroups of one variant that fit a CU together (registers and LDS), asked from the runtime
// once per variant and LDS size; 3 when there is no device to ask.
template <typename T, int NT>
int fwd_resident_k(int CPG, size_t lds) {
  const void* k = nullptr;
  switch (CPG) {
    case 1: k = (const void*)conv3x3_mfma_kernel<T, NT, 1, false>; break;
    case 2: k = (const void*)conv3x3_mfma_kernel<T, NT, 2, false>; break;
    case 3: k = (const void*)conv3x3_mfma_kernel<T, NT, 3, false>; break;
    default: k = (const void*)conv3x3_mfma_kernel<T, NT, 4, false>; break;
  }
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, CM_THREADS, lds) != hipSuccess || n < 1) {
    (void)hipGetLastError();
    return 3;
  }
  return n;
}

template <typename T>
int fwd_resident(int NT, int CPG, size_t lds) {
  static int cache[5][5][8] = {};                 // [NT][CPG][lds / 32 KiB]
  int& c = cache[NT][CPG][(lds >> 15) & 7];
  if (c == 0) c = NT == 4 ? fwd_resident_k<T, 4>(CPG, lds) : NT == 2 ? fwd_resident_k<T, 2>(CPG, lds)
                                                                     : fwd_resident_k<T, 1>(CPG, lds);
  return c;
}

// Largest problem (16x16-pixel tiles) that takes the wave-specialised form: beyond it the layers are
// bandwidth-bound and several small workgroups per CU keep more bytes in flight than one large one
// (measured on the 512^2 layers of A_UNet / RDAM_UNet: 16384 tiles; the 64-channel layers at 256^2,
// 4096 tiles, gain from the form: 64->64 121 -> 102 us).
constexpr int WS_MAX_TILES = 4096;

// Workgroup counts forced through mis_set_conv_launch (0 = sized by the planners below).
int g_fwd_workgroups = 0, g_wgrad_workgroups = 0;

// Ca_in: channels of the first input tensor (== Cin for one tensor); Cout_a: channels of the
// first output tensor (== Cout for one tensor).
template <typename T>
bool plan_fwd(int B, int H, int W, int Cin, int Cout, FwdPlan& p, int Ca_in = -1, int Cout_a = -1, int dil = 1) {
  constexpr int G = Mf<T>::G;
  if (Cin % G != 0 || Cout % 4 != 0 || Cin < G) return false;
  if ((int64_t)Cin * Cout >= ((int64_t)1 << 26)) return false;      // packed image indices stay below 2^31
  // one image of either tensor is addressed through a 32-bit buffer descriptor (byte offsets, and
  // 0x7FFFFFFF means "out of range"): larger images take the direct kernels
  if ((int64_t)H * W * (Cin > Cout ? Cin : Cout) * (int64_t)sizeof(T) >= ((int64_t)1 << 31) - (1 << 22)) return false;
  const int groups = Cin / G;
  if (groups <= 4) {
    p.CPG = groups;
    p.n_chunks = 1;
  } else if (groups % 4 == 0) {
    p.CPG = 4;
    p.n_chunks = groups / 4;
  } else {
    return false;
  }
  if (Ca_in >= 0 && Ca_in != Cin) {
    // two input tensors: equal channel counts, each a whole number of K chunks
    if (p.CPG != 4 || 2 * Ca_in != Cin || Ca_in % (4 * G) != 0) return false;
  }
  p.nt_total = (Cout + 15) / 16;
  p.NT = p.nt_total >= 4 ? 4 : (p.nt_total >= 2 ? 2 : 1);
  if (p.nt_total % p.NT != 0) p.NT = (p.nt_total % 2 == 0) ? 2 : 1;
  if (sizeof(T) == 2 && Cout % 8 != 0) p.NT = 1;      // the paired 16-byte stores need whole 8-channel groups
  if (Cout_a >= 0 && Cout_a != Cout) {
    // two output tensors: a workgroup's 16*NT channel slice must lie inside one of them
    if (Cout_a % 16 != 0 || Cout_a <= 0 || Cout_a >= Cout) return false;
    while ((Cout_a / 16) % p.NT != 0) p.NT /= 2;
    if (sizeof(T) == 2 && (Cout - Cout_a) % 8 != 0) p.NT = 1;
  }
  const int tiles16 = B * ((W + 15) / 16) * ((H + 15) / 16);
  p.ws = p.CPG == 4 && p.NT >= 2 && dil == 1 && tiles16 <= WS_MAX_TILES;
  if (!p.ws) {
    // few pixels (the center layers): 64-channel slices would leave one workgroup per CU; 32-channel
    // slices double the workgroups, and two resident per CU hide each other's staging (measured
    // 15-25 % on those layers; with >= 2 workgroups per CU already, the wider slice is faster)
    if (p.NT == 4 && tiles16 * (p.nt_total / 4) < 512) p.NT = 2;
  }
  p.grid_y = p.nt_total / p.NT;
  const int MX = fwd_mx(p.NT, p.CPG), TW = 16 * MX, TH = 4 * (fwd_mt(p.NT, p.CPG) / MX), HW_ = TW + 2;
  p.tiles_x = (W + TW - 1) / TW;
  p.tiles_y = (H + TH - 1) / TH;
  p.n_tiles = B * p.tiles_x * p.tiles_y;
  p.lds = (p.ws ? 2 : 1) * ((size_t)(TH + 2 * dil) * (TW + 2 * dil) * (p.CPG == 4 ? 96 : (p.CPG | 1) * 16) + 16 +
                            (size_t)nsteps_of(p.CPG) * p.NT * 1024) +
          (size_t)p.NT * 16 * 4 + 2 * (size_t)Cin * 4 + 4 * (size_t)p.NT * 16 * 4;
  if (p.ws && p.lds > 160 * 1024) return false;
  // Total workgroups per launch: as many as are resident at once on 256 CUs, so the persistent
  // workgroups run as one round (a second, partly filled round costs a whole tile range: 10-25 %
  // on the 128^2 and 256^2 layers).
  // (The 8-channel-input variants stream many tiles per workgroup and measured best at 768 whatever
  // their residency: shorter ranges balance better than one round of long ones.)
  const int target = p.ws ? 256 : p.CPG == 1 ? 768 : 256 * fwd_resident<T>(p.NT, p.CPG, p.lds);
  int gx = target / p.grid_y;
  if (g_fwd_workgroups > 0) {
    gx = g_fwd_workgroups;                 // mis_set_conv_launch: tests walk many tiles per workgroup
  } else if (p.ws) {
    // one workgroup per CU, all resident at once: ceil so that no CU is left without one
    gx = (target + p.grid_y - 1) / p.grid_y;
    if (gx < 1) gx = 1;
  } else if (gx < 256) {
    gx = 256;
  }
  if (gx > p.n_tiles) gx = p.n_tiles;
  p.grid_x = gx;
  return true;
}

struct FwdArgs {
  ConvIn in;
  ConvOut out;
  const void* wp;
  const float* bias;
  float* partials;
  int B, H, W;
  const void* bnz;          // BatchNorm-backward sums of relu(bn(bnz)) with the output as gradient (or null)
  const float* bn_aff;      // its [4][Cout] (scale, shift, mean, rstd)
  int dil;                  // dilation of the taps: 1, 2 or 3
};

template <typename T, int NT, int CPG>
void launch_fwd_k(const FwdPlan& p, const FwdArgs& a, hipStream_t st) {
  auto kfn = a.bnz ? conv3x3_mfma_kernel<T, NT, CPG, true, 1>
             : a.dil == 2 ? conv3x3_mfma_kernel<T, NT, CPG, false, 2>
             : a.dil == 3 ? conv3x3_mfma_kernel<T, NT, CPG, false, 3>
                          : conv3x3_mfma_kernel<T, NT, CPG, false, 1>;
  if constexpr (CPG == 4) {
    if (a.dil == 0 && !a.bnz) kfn = conv3x3_mfma_kernel<T, NT, 4, false, 0>;     // centre tap only
  }
  int threads = CM_THREADS;
  if constexpr (CPG == 4 && NT >= 2) {
    if (p.ws) {
      kfn = a.bnz ? conv3x3_mfma_kernel<T, NT, 4, true, 1, true> : conv3x3_mfma_kernel<T, NT, 4, false, 1, true>;
      threads = 2 * CM_THREADS;
    }
  }
  if (p.lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
  const int Cin = a.in.channels(), Cout = a.out.channels();
  const int n_chunks_a = a.in.xb ? a.in.Ca / (4 * Mf<T>::G) : p.n_chunks;
  kfn<<<dim3(p.grid_x, p.grid_y), threads, p.lds, st>>>(
      (const T*)a.in.xa, (const T*)a.in.xb, (const T*)a.wp, a.bias, (T*)a.out.za, (T*)a.out.zb, a.partials, a.B, a.H,
      a.W, Cin, a.in.Ca, n_chunks_a, Cout, a.out.Ca, p.n_chunks, p.nt_total, p.tiles_x, p.tiles_y, p.n_tiles,
      a.in.affa, a.in.affb, (const T*)a.bnz, a.bn_aff);
}

template <typename T, int NT>
void launch_fwd_cpg(const FwdPlan& p, const FwdArgs& a, hipStream_t st) {
  switch (p.CPG) {
    case 1: launch_fwd_k<T, NT, 1>(p, a, st); break;
    case 2: launch_fwd_k<T, NT, 2>(p, a, st); break;
    case 3: launch_fwd_k<T, NT, 3>(p, a, st); break;
    default: launch_fwd_k<T, NT, 4>(p, a, st); break;
  }
}

template <typename T>
int run_fwd(const ConvIn& in, const void* wp, const float* bias, int B, int H, int W, const ConvOut& out,
            float* stats, void* workspace, hipStream_t st, const void* bnz = nullptr, const float* bn_aff = nullptr,
            int dil = 1) {
  FwdPlan p;
  const int Cin = in.channels(), Cout = out.channels();
  if (dil < 0 || dil > 3 || (dil != 1 && bnz)) {
    mis_set_error("conv3x3_mfma: dilation %d not supported here (0..3; the fused BatchNorm sums only with 1)", dil);
    return MIS_ERR_UNSUPPORTED;
  }
  // dil == 0: the caller states that only the centre tap is non-zero; shapes whose K steps span several taps run
  // the full 3x3 (same result)
  if (dil == 0 && !(plan_fwd<T>(B, H, W, Cin, Cout, p, in.Ca, out.Ca, 0) && p.CPG == 4)) dil = 1;
  if (!plan_fwd<T>(B, H, W, Cin, Cout, p, in.Ca, out.Ca, dil)) {
    mis_set_error("conv3x3_mfma: unsupported shape Cin=%d+%d Cout=%d+%d", in.Ca, in.Cb, out.Ca, out.Cb);
    return MIS_ERR_UNSUPPORTED;
  }
  FwdArgs a{in, out, wp, bias, stats ? (float*)workspace : nullptr, B, H, W, bnz, bn_aff, dil};
  switch (p.NT) {
    case 1: launch_fwd_cpg<T, 1>(p, a, st); break;
    case 2: launch_fwd_cpg<T, 2>(p, a, st); break;
    default: launch_fwd_cpg<T, 4>(p, a, st); break;
  }
  MIS_CHECK_LAUNCH("conv3x3_mfma_kernel");
  if (stats && (void*)stats != workspace) {      // stats == workspace: the caller reduces the partial rows itself
    launch_reduce_partials(a.partials, p.grid_x, 2 * Cout, stats, st);
    MIS_CHECK_LAUNCH("reduce_partials_kernel");
  }
  return MIS_OK;
}

}  // namespace

bool conv3x3_mfma_supported(int dtype, int B, int H, int W, int Cin, int Cout) {
  FwdPlan p;
  return dtype == MIS_BF16 ? plan_fwd<bf16>(B, H, W, Cin, Cout, p) : plan_fwd<float>(B, H, W, Cin, Cout, p);
}

bool conv3x3_mfma_cat_supported(int dtype, int Ca, int Cb, int Cout) {
  FwdPlan p;
  if (Ca != Cb) return false;
  return dtype == MIS_BF16 ? plan_fwd<bf16>(1, 16, 16, Ca + Cb, Cout, p, Ca, -1)
                           : plan_fwd<float>(1, 16, 16, Ca + Cb, Cout, p, Ca, -1);
}

bool conv3x3_mfma_split_supported(int dtype, int Cin, int Ca, int Cb) {
  FwdPlan p;
  return dtype == MIS_BF16 ? plan_fwd<bf16>(1, 16, 16, Cin, Ca + Cb, p, -1, Ca)
                           : plan_fwd<float>(1, 16, 16, Cin, Ca + Cb, p, -1, Ca);
}

int conv3x3_mfma_stats_rows(int dtype, int B, int H, int W, int Cin, int Cout, int dil) {
  FwdPlan p;
  bool ok = dtype == MIS_BF16 ? plan_fwd<bf16>(B, H, W, Cin, Cout, p, -1, -1, dil)
                              : plan_fwd<float>(B, H, W, Cin, Cout, p, -1, -1, dil);
  if (dil == 0 && !(ok && p.CPG == 4))       // (the centre-tap form falls back to the full 3x3, as run_fwd does)
    ok = dtype == MIS_BF16 ? plan_fwd<bf16>(B, H, W, Cin, Cout, p) : plan_fwd<float>(B, H, W, Cin, Cout, p);
  return ok ? p.grid_x : 0;
}

size_t conv3x3_mfma_workspace_bytes(int dtype, int B, int H, int W, int Cin, int Cout) {
  return align256((size_t)8192 * 2 * Cout * sizeof(float));
}

size_t conv3x3_mfma_packed_bytes(int dtype, int Cin, int Cout) {
  const int G = dtype == MIS_BF16 ? 8 : 4;
  const int groups = Cin / G;
  const int CPG = groups <= 4 ? groups : 4;
  const int n_chunks = groups <= 4 ? 1 : groups / 4;
  return (size_t)n_chunks * nsteps_of(CPG) * ((Cout + 15) / 16) * 1024;
}

namespace {
struct PackGeom {
  int CPG, n_chunks, nt_total;
  int64_t total;
};
PackGeom pack_geom(int Cin, int Cout, int dtype) {
  const int G = dtype == MIS_BF16 ? 8 : 4;
  const int groups = Cin / G;
  PackGeom g;
  g.CPG = groups <= 4 ? groups : 4;
  g.n_chunks = groups <= 4 ? 1 : groups / 4;
  g.nt_total = (Cout + 15) / 16;
  g.total = (int64_t)g.n_chunks * nsteps_of(g.CPG) * g.nt_total * 64 * G;
  return g;
}
}  // namespace

int conv3x3_mfma_pack(const float* w, int Cw_out, int Cw_in, int transposed, int dtype, void* out, hipStream_t st) {
  const int Cin = transposed ? Cw_out : Cw_in, Cout = transposed ? Cw_in : Cw_out;   // conv roles
  const PackGeom g = pack_geom(Cin, Cout, dtype);
  const int grid = ew_grid(g.total, 256);
  if (dtype == MIS_BF16)
    pack_w3x3_frag_kernel<bf16><<<grid, 256, 0, st>>>(w, Cw_out, Cw_in, transposed, g.CPG, g.n_chunks, g.nt_total,
                                                       (bf16*)out, g.total, 0, 0, 0, (bf16*)nullptr, 0);
  else
    pack_w3x3_frag_kernel<float><<<grid, 256, 0, st>>>(w, Cw_out, Cw_in, transposed, g.CPG, g.n_chunks, g.nt_total,
                                                        (float*)out, g.total, 0, 0, 0, (float*)nullptr, 0);
  MIS_CHECK_LAUNCH("pack_w3x3_frag_kernel");
  return MIS_OK;
}

int conv3x3_mfma_pack_pair(const float* w, int Cw_out, int Cw_in, int dtype, void* out_fwd, void* out_t,
                           hipStream_t st) {
  const PackGeom f = pack_geom(Cw_in, Cw_out, dtype), t = pack_geom(Cw_out, Cw_in, dtype);
  const int grid = ew_grid(f.total + t.total, 256);
  if (dtype == MIS_BF16)
    pack_w3x3_frag_kernel<bf16><<<grid, 256, 0, st>>>(w, Cw_out, Cw_in, 0, f.CPG, f.n_chunks, f.nt_total,
                                                       (bf16*)out_fwd, f.total, t.CPG, t.n_chunks, t.nt_total,
                                                       (bf16*)out_t, t.total);
  else
    pack_w3x3_frag_kernel<float><<<grid, 256, 0, st>>>(w, Cw_out, Cw_in, 0, f.CPG, f.n_chunks, f.nt_total,
                                                        (float*)out_fwd, f.total, t.CPG, t.n_chunks, t.nt_total,
                                                        (float*)out_t, t.total);
  MIS_CHECK_LAUNCH("pack_w3x3_frag_kernel");
  return MIS_OK;
}

int conv3x3_mfma_pack_job(const float* w, int Cw_out, int Cw_in, int dtype, void* out_fwd, void* out_t,
                          int first_block, mis_pack_job* job) {
  const PackGeom f = pack_geom(Cw_in, Cw_out, dtype), t = pack_geom(Cw_out, Cw_in, dtype);
  const int64_t G = dtype == MIS_BF16 ? Mf<bf16>::G : Mf<float>::G;
  int nb = (int)cdiv((f.total + t.total) / G, (int64_t)256);      // one 16-byte vector per thread
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  *job = mis_pack_job{w, out_fwd, out_t, f.total, t.total, Cw_out, Cw_in, f.CPG, f.nt_total, t.CPG, t.nt_total,
                      first_block, nb};
  return nb;
}

int conv3x3_mfma_pack_multi(const mis_pack_job* jobs_dev, int n_jobs, int total_blocks, int dtype, hipStream_t st) {
  if (dtype == MIS_BF16) pack_w3x3_multi_kernel<bf16><<<total_blocks, 256, 0, st>>>(jobs_dev, n_jobs);
  else pack_w3x3_multi_kernel<float><<<total_blocks, 256, 0, st>>>(jobs_dev, n_jobs);
  MIS_CHECK_LAUNCH("pack_w3x3_multi_kernel");
  return MIS_OK;
}

int conv3x3_mfma_fwd(const ConvIn& in, const void* wp, const float* bias, int dtype, int B, int H, int W,
                     const ConvOut& out, float* stats, void* workspace, hipStream_t st, const void* bnz,
                     const float* bn_aff, int dil) {
  if (dtype == MIS_BF16) return run_fwd<bf16>(in, wp, bias, B, H, W, out, stats, workspace, st, bnz, bn_aff, dil);
  return run_fwd<float>(in, wp, bias, B, H, W, out, stats, workspace, st, bnz, bn_aff, dil);
}

// =====================================================================================
// Weight gradient on the matrix cores.
//
//   dW[tap][co][ci] = sum over pixels p of dz[p][co] * x[p + off(tap)][ci]
//
// A GEMM whose K dimension is the pixel index: both operands are "K-major" in a
// channels-last tensor (a lane needs 8 consecutive PIXELS of one channel), which is exactly
// what ds_read_b64_tr_b16 delivers from a row-major [pixel][channel] LDS tile: no software
// transpose.  A workgroup stages a TH x 32 pixel tile of dz and the matching (TH+2) x 34 halo
// of x, each of its K-steps is one 32-pixel row segment, and the dz fragment of a step is
// reused by the 9 taps (the x fragment is the same row shifted by the tap).  Workgroups walk
// tiles grid-stride and keep the 9 x (co-tile x ci-tile) accumulators in registers; at the
// end each writes its partial to a slab, and a second kernel sums the slabs in a fixed order
// (bitwise reproducible, no float atomics) into the nn.Conv2d layout [Cout][Cin][3][3].
//
// Wave roles (template): WCO x WCI x WK = 4 waves.  Small channel counts: all waves split
// the rows of a tile (WK = 4) and are summed through LDS at the end; large: waves tile a
// 64 x 64 (co x ci) block.
// float32 mode uses v_mfma_f32_16x16x4_f32 with plain 4-byte LDS reads (K-step = 4 pixels).
//
// Roofline: min(MFMA, HBM); algorithmic bytes = B*H*W*(Cin+Cout)*sizeof(T) + 36*Cin*Cout.
namespace {

constexpr int WG_TW = 32;

template <typename T> struct WgT;
template <> struct WgT<bf16> { static constexpr int TH = 8; };
template <> struct WgT<float> { static constexpr int TH = 4; };
// Tile height of a variant: the wave-specialised form holds two stages in LDS, so its 64x64-channel
// block takes 4-row tiles.
template <typename T>
constexpr int wg_tile_h(bool ws, int cto, int cti) { return (ws && cto * cti >= 4096) ? 4 : WgT<T>::TH; }
constexpr int WG_PRODUCERS = 256;      // staging threads of the wave-specialised form

// LDS tiles are [position][CT channels] with rows padded to an odd multiple of 32 bytes, so
// that the 8 consecutive positions one half-wave transposed read touches fall on 8 distinct
// 32-byte bank windows (conflict-free) while every address stays "per-lane base + uniform
// offset" (tap shifts fold into the instruction's immediate offset).
constexpr int padded_row_bytes(int row_bytes) { return ((row_bytes / 32) % 2 == 0 && row_bytes > 32) ? row_bytes + 32 : row_bytes; }

// K index of the MFMA (k = 8g + 4h + q; g = lane group, h = which of the two transposed
// reads, q = row inside the read) -> pixel offset inside the 32-pixel step.  Chosen so that
// the two groups of a half-wave read 8 CONSECUTIVE pixels.
__device__ __forceinline__ int k_pixel(int g, int h, int q) { return 16 * (g >> 1) + 8 * h + 4 * (g & 1) + q; }

// 32 pixels x 16 channels -> lane i of group g: channel c0+i, its 8 k values.  `lane_off` is
// the lane's byte offset (k_pixel(g,0,q)*RSP + 8p), `uni_off` the wave-uniform byte offset of
// (first position, first channel); the second read is 8 positions further.
template <int RSP>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int lane_off, int uni_off) {
  typedef __attribute__((address_space(3))) bf16x4 lds_v4;
  const unsigned char* a0 = tile + lane_off + uni_off;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)a0);
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(a0 + 8 * RSP));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// The four-wave (light-channel) variants are bound by memory latency: three resident workgroups per
// CU (<= 168 registers) keep a third more bytes in flight than two and run 15-25% faster.
//
// WS (wave-specialised): 4 extra producer waves stage tile t+1 (global -> registers -> affine -> LDS)
// into the second of two LDS stages while the MFMA waves work on tile t; one barrier per tile.
// In the single-role form the staging and the MFMAs of a tile alternate (measured: 45 % of a
// tile's time outside the MFMA phase on the deep layers).
template <typename T, int MTO, int NTI, int WCO, int WCI, int WK, int DIL = 1, bool WS = false>
__global__ __launch_bounds__(64 * WCO * WCI * WK + (WS ? WG_PRODUCERS : 0), (WS ? 1 : WK == 4 ? 3 : 1)) void conv3x3_wgrad_mfma_kernel(
    const T* __restrict__ xa, const T* __restrict__ xb, int Ca, const T* __restrict__ dz, float* __restrict__ slab,
    int B, int H, int W, int Cin, int Cout, int n_ci_blocks, int tiles_x, int tiles_y, int n_tiles,
    const float* __restrict__ affa, const float* __restrict__ affb, int nsplit, int nout) {
  constexpr int NTHR = 64 * WCO * WCI * WK;            // MFMA threads
  constexpr int STHR = WS ? WG_PRODUCERS : NTHR;       // staging threads
  constexpr int G = Mf<T>::G;
  constexpr int TH = wg_tile_h<T>(WS, WCO * MTO * 16, WCI * NTI * 16);
  // DIL == 0: the layer is a 1x1 conv carried as a 3x3 — only the centre tap of its gradient is wanted: no halo, one
  // tap multiplied (the other eight accumulators stay zero and are written as zeros, so the reduction is unchanged)
  constexpr int WT0 = DIL == 0 ? 4 : 0, WT1 = DIL == 0 ? 5 : 9;
  constexpr int HWW = WG_TW + 2 * DIL;           // halo width (34 without dilation)
  constexpr int PH = (TH + 2 * DIL) * HWW;       // halo positions
  constexpr int PZ = TH * WG_TW;                 // dz positions
  constexpr int CTO = WCO * MTO * 16, CTI = WCI * NTI * 16;
  constexpr int XVP = CTI / G, ZVP = CTO / G;    // 16-byte vectors per position
  constexpr int XV = PH * XVP, ZV = PZ * ZVP;
  constexpr int XIT = (XV + STHR - 1) / STHR, ZIT = (ZV + STHR - 1) / STHR;
  constexpr int RSX = padded_row_bytes(CTI * (int)sizeof(T)), RSZ = padded_row_bytes(CTO * (int)sizeof(T));
  constexpr int STAGE_BYTES = PH * RSX + 16 + PZ * RSZ + 16;   // one stage: x halo + dz tile (+ dummy slots)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xh = smem;                      // [PH] rows of RSX bytes (CTI channels + pad), + 1 dummy slot
  unsigned char* zt = smem + PH * RSX + 16;      // [PZ] rows of RSZ bytes, + 1 dummy slot
  float* aff = reinterpret_cast<float*>(smem + (WS ? 2 : 1) * STAGE_BYTES);   // [2][CTI] (scale, shift) of this block's x channels

  const bool producer = WS && threadIdx.x >= NTHR;
  // `tid` numbers the staging threads where the staging code uses it (WS: the producers, 0..255)
  const int tid = producer ? (int)threadIdx.x - NTHR : (int)threadIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wk = producer ? WK : wid % WK, wci = (wid / WK) % WCI, wco = producer ? 0 : wid / (WK * WCI);
  // One-dimensional grid of nsplit * nout workgroups.  The nout (co, ci) blocks of one tile range
  // read the same x and dz tiles; consecutive workgroup ids go round the 8 XCDs (each with its own
  // L2), so with nsplit a multiple of 8 the blocks of a range are given ids 8 apart: same XCD, launched
  // together, and the second reader of a cache line finds it in that L2.
  int bx, by;
  {
    const int id = blockIdx.x;
    if ((nsplit & 7) == 0) {
      const int j = id >> 3;
      by = j % nout;
      bx = (j / nout) * 8 + (id & 7);
    } else {
      by = id % nout;
      bx = id / nout;
    }
  }
  const int cib = by % n_ci_blocks, cob = by / n_ci_blocks;
  const int ci0 = cib * CTI, co0 = cob * CTO;
  // two-tensor input [xa (Ca channels) | xb]: this workgroup's channel tile lies in one of them
  const bool sb = xb != nullptr && ci0 >= Ca;
  const T* const x = sb ? xb : xa;
  const float* const x_aff = sb ? affb : affa;
  const int Csrc = xb == nullptr ? Cin : (sb ? Cin - Ca : Ca);     // channels (= pixel stride) of that tensor
  const int cl0 = sb ? ci0 - Ca : ci0;                              // first channel of the tile inside it

  f32x4 acc[9][MTO][NTI];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int mo = 0; mo < MTO; mo++)
#pragma unroll
      for (int ni = 0; ni < NTI; ni++) acc[t][mo][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Tile-invariant staging state of this lane's 16-byte elements (x halo and dz tile): byte
  // offset relative to the tile origin, column inside the tile (huge for tail / channel-tail
  // elements: never loaded), LDS address (a dummy slot for those).  Per tile a load then costs an
  // add, a compare and a select.  Rows above/below the image fall outside the per-image buffer
  // descriptor by themselves and read zeros; only the column needs the compare.
  constexpr unsigned OOB = 0x7FFFFFFFu;
  unsigned x_rel[XIT], z_rel[ZIT];
  int x_px[XIT], z_px[ZIT], x_lds[XIT], z_lds[ZIT];
#pragma unroll
  for (int i = 0; i < XIT; i++) {
    const int v = tid + i * STHR, pos = v / XVP, cv = v % XVP;
    const bool valid = v < XV && cl0 + cv * G < Csrc;
    x_rel[i] = (unsigned)((((pos / HWW) * W + pos % HWW) * Csrc + cl0 + cv * G) * (int)sizeof(T));
    x_px[i] = valid ? pos % HWW : (1 << 28);
    x_lds[i] = valid ? pos * RSX + cv * 16 : PH * RSX;
  }
#pragma unroll
  for (int i = 0; i < ZIT; i++) {
    const int v = tid + i * STHR, pos = v / ZVP, cv = v % ZVP;
    const bool valid = v < ZV && co0 + cv * G < Cout;
    z_rel[i] = (unsigned)((((pos / WG_TW) * W + pos % WG_TW) * Cout + co0 + cv * G) * (int)sizeof(T));
    z_px[i] = valid ? pos % WG_TW : (1 << 28);
    z_lds[i] = valid ? pos * RSZ + cv * 16 : PZ * RSZ;
  }

  u32x4 xr[XIT], zr[ZIT];
  // (the variants at their register limit would spill the 16 extra values: they keep the per-element form)
  constexpr bool WG_AFF_REGS = (STHR % XVP == 0) && (MIS_AFF_REGS != 0) && !(NTI == 2 && (WK == 4 || WCO == 2));
  const int ximg_bytes = H * W * Csrc * (int)sizeof(T), zimg_bytes = H * W * Cout * (int)sizeof(T);
  // a dead stage (past the last tile) shifts x far left: nothing loads
#define WG_LOAD(P, LIVE)                                                                                \
  {                                                                                                     \
    const int y0_ = (P).ty * TH, xo_ = (P).tx * WG_TW;                                                  \
    const int x0_ = (LIVE) ? xo_ : -(1 << 29);                                                          \
    const __amdgpu_buffer_rsrc_t xrs_ = __builtin_amdgcn_make_buffer_rsrc(                              \
        (void*)(x + (int64_t)(P).b * H * W * Csrc), 0, ximg_bytes, 0x00020000);                         \
    const __amdgpu_buffer_rsrc_t zrs_ = __builtin_amdgcn_make_buffer_rsrc(                              \
        (void*)(dz + (int64_t)(P).b * H * W * Cout), 0, zimg_bytes, 0x00020000);                        \
    const unsigned xb_ = (unsigned)((((y0_ - DIL) * W + xo_ - DIL) * Csrc) * (int)sizeof(T));           \
    const unsigned zb_ = (unsigned)(((y0_ * W + xo_) * Cout) * (int)sizeof(T));                         \
    _Pragma("unroll") for (int i = 0; i < XIT; i++) {                                                   \
      const bool ok = (unsigned)(x0_ - DIL + x_px[i]) < (unsigned)W;                                    \
      xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs_, (int)(ok ? xb_ + x_rel[i] : OOB), 0, 0);      \
    }                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < ZIT; i++) {                                                   \
      const bool ok = (unsigned)(x0_ + z_px[i]) < (unsigned)W;                                          \
      zr[i] = __builtin_amdgcn_raw_buffer_load_b128(zrs_, (int)(ok ? zb_ + z_rel[i] : OOB), 0, 0);      \
    }                                                                                                   \
  }
#define WG_STORE(P)                                                                                     \
  {                                                                                                     \
    if (WG_AFF_REGS && x_aff) {                                                                         \
      /* a thread's elements all have the same channel vector (NTHR is a multiple of XVP): scale and */ \
      /* shift are read once per stage, the in-image test is a select (no per-element LDS round trip) */ \
      const int cv_ = tid % XVP;                                                                        \
      float sc_[G], sh_[G];                                                                             \
      _Pragma("unroll") for (int k = 0; k < G; k += 4) {                                                \
        const f32x4 a_ = *reinterpret_cast<const f32x4*>(aff + cv_ * G + k);                            \
        const f32x4 b_ = *reinterpret_cast<const f32x4*>(aff + CTI + cv_ * G + k);                      \
        _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                 \
          sc_[k + r] = a_[r];                                                                           \
          sh_[k + r] = b_[r];                                                                           \
        }                                                                                               \
      }                                                                                                 \
      const int gy0_ = (P).ty * TH - DIL, gx0_ = (P).tx * WG_TW - DIL;                                  \
      _Pragma("unroll") for (int i = 0; i < XIT; i++) {                                                 \
        const u32x4 raw_ = xr[i];                                                                       \
        const unsigned pos = ((unsigned)tid + i * STHR) / XVP;                                          \
        const bool in_ = (unsigned)(gy0_ + (int)(pos / HWW)) < (unsigned)H &&                           \
                         (unsigned)(gx0_ + (int)(pos % HWW)) < (unsigned)W;                             \
        const u32x4 act_ = affine_relu_regs<T, G>(raw_, sc_, sh_);                                      \
        u32x4 val;                                                                                      \
        _Pragma("unroll") for (int r = 0; r < 4; r++) val[r] = in_ ? act_[r] : raw_[r];                 \
        *reinterpret_cast<u32x4*>(xh + x_lds[i]) = val;                                                 \
      }                                                                                                 \
    } else {                                                                                            \
      _Pragma("unroll") for (int i = 0; i < XIT; i++) {                                                 \
        u32x4 val = xr[i];                                                                              \
        if (x_aff) {                                                                                    \
          const int v = tid + i * STHR, pos = v / XVP, cv = v % XVP;                                    \
          const int gy = (P).ty * TH - DIL + pos / HWW, gx = (P).tx * WG_TW - DIL + pos % HWW;          \
          if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && cl0 + cv * G < Csrc)          \
            val = affine_relu_raw<T>(val, aff + cv * G, aff + CTI + cv * G);                            \
        }                                                                                               \
        *reinterpret_cast<u32x4*>(xh + x_lds[i]) = val;                                                 \
      }                                                                                                 \
    }                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < ZIT; i++) *reinterpret_cast<u32x4*>(zt + z_lds[i]) = zr[i];   \
  }

  // per-lane byte offsets of the transposed reads (bf16) inside the two tiles
  const int lg = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const int offx = k_pixel(lg, 0, lq) * RSX + 8 * lp + wci * NTI * 32;
  const int offz = k_pixel(lg, 0, lq) * RSZ + 8 * lp + wco * MTO * 32;

  if (x_aff) {
    for (int i = threadIdx.x; i < 2 * CTI; i += blockDim.x) {
      const int c = cl0 + i % CTI;
      aff[i] = (c < Csrc) ? x_aff[(i / CTI) * Csrc + c] : 0.f;
    }
  }
  // this workgroup's contiguous tile range (x fastest, then y, then image): coordinates advance
  // by carries, neighbouring halos overlap in the same CU's cache
  const int nb_ = nsplit, q_ = n_tiles / nb_, r_ = n_tiles % nb_;
  const int t_begin = bx * q_ + (bx < r_ ? bx : r_);
  const int t_count = q_ + (bx < r_ ? 1 : 0);
  TilePos cur;
  cur.tx = t_begin % tiles_x;
  cur.ty = (t_begin / tiles_x) % tiles_y;
  cur.b = t_begin / (tiles_x * tiles_y);
#if MIS_TIMELINE
  int tl_n = 0;
  const int tl_slot = (tid != 0) ? -1
                      : WS ? (blockIdx.x == 0 ? (producer ? 1 : 0) : (blockIdx.x == gridDim.x - 1 && !producer) ? 2 : -1)
                      : blockIdx.x == 0 ? 0 : blockIdx.x == gridDim.x / 2 ? 1 : blockIdx.x == gridDim.x - 1 ? 2 : -1;
  TL_STAMP(2)
#endif
#define WG_MFMA_ROWS() \
  { \
    for (int r = wk; r < TH; r += WK) { \
      if constexpr (sizeof(T) == 2) { \
        bf16x8 a[MTO]; \
    _Pragma("unroll") \
        for (int mo = 0; mo < MTO; mo++) a[mo] = tr_frag<RSZ>(zt, offz, r * WG_TW * RSZ + mo * 32); \
    _Pragma("unroll") \
        for (int t = WT0; t < WT1; t++) { \
    _Pragma("unroll") \
          for (int ni = 0; ni < NTI; ni++) { \
            const bf16x8 bq = tr_frag<RSX>(xh, offx, (r * HWW + ((t / 3) * HWW + (t % 3)) * DIL) * RSX + ni * 32); \
    _Pragma("unroll") \
            for (int mo = 0; mo < MTO; mo++) \
              acc[t][mo][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mo], bq, acc[t][mo][ni], 0, 0, 0); \
          } \
        } \
      } else { \
    _Pragma("unroll 2") \
        for (int kk = 0; kk < WG_TW / 4; kk++) { \
          const int px = 4 * kk + (lane >> 4); \
          float a[MTO]; \
    _Pragma("unroll") \
          for (int mo = 0; mo < MTO; mo++) { \
            const int pos = r * WG_TW + px, c = (wco * MTO + mo) * 16 + (lane & 15); \
            a[mo] = *reinterpret_cast<const float*>(zt + pos * RSZ + c * 4); \
          } \
    _Pragma("unroll") \
          for (int t = WT0; t < WT1; t++) { \
    _Pragma("unroll") \
            for (int ni = 0; ni < NTI; ni++) { \
              const int pos = (r + (t / 3) * DIL) * HWW + (t % 3) * DIL + px, c = (wci * NTI + ni) * 16 + (lane & 15); \
              const float bq = *reinterpret_cast<const float*>(xh + pos * RSX + c * 4); \
    _Pragma("unroll") \
              for (int mo = 0; mo < MTO; mo++) \
                acc[t][mo][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mo], bq, acc[t][mo][ni], 0, 0, 0); \
            } \
          } \
        } \
      } \
    } \
  }
  if constexpr (WS) {
#define WG_SET_BUF(BUFI)                                                                                \
  {                                                                                                     \
    xh = smem + (BUFI) * STAGE_BYTES;                                                                   \
    zt = xh + PH * RSX + 16;                                                                            \
  }
    if (producer) {
      WG_LOAD(cur, t_count > 0)           // (requested before the affine table is visible)
      __syncthreads();
      WG_SET_BUF(0)
      WG_STORE(cur)
      cur.next(tiles_x, tiles_y);
      WG_LOAD(cur, 1 < t_count)
      __syncthreads();
      for (int it = 0; it < t_count; it++) {
        TL_STAMP(10)
        if (it + 1 < t_count) {
          TL_WAIT_LOADS()
          WG_SET_BUF((it + 1) & 1)
          WG_STORE(cur)
        }
        TL_STAMP(12)
        cur.next(tiles_x, tiles_y);
        WG_LOAD(cur, it + 2 < t_count)
        TL_STAMP(14)
        __syncthreads();
      }
    } else {
      __syncthreads();
      __syncthreads();
      for (int it = 0; it < t_count; it++) {
        TL_STAMP(14)
        WG_SET_BUF(it & 1)
        WG_MFMA_ROWS()
        TL_STAMP(15)
        __syncthreads();
        TL_STAMP(13)
      }
    }
#undef WG_SET_BUF
  } else {
  WG_LOAD(cur, t_count > 0)
  for (int it = 0; it < t_count; it++) {
    TilePos nxt = cur;
    nxt.next(tiles_x, tiles_y);
    TL_STAMP(10)
    __syncthreads();
    TL_STAMP(11)
    TL_WAIT_LOADS()
    WG_STORE(cur)
    TL_STAMP(12)
    __syncthreads();
    TL_STAMP(13)
    WG_LOAD(nxt, it + 1 < t_count)
    __builtin_amdgcn_sched_barrier(0);     // the loads stay ahead of the MFMAs
    TL_STAMP(14)

    WG_MFMA_ROWS()
    TL_STAMP(15)
    cur = nxt;
  }
  }
#undef WG_MFMA_ROWS
#undef WG_LOAD
#undef WG_STORE
  TL_STAMP(20)

  // sum the WK row-split waves through LDS (wave wk=0 of each (wco,wci) keeps the total)
  if constexpr (WK > 1) {
    f32x4* red = reinterpret_cast<f32x4*>(smem) + (wco * WCI + wci) * (9 * MTO * NTI * 64);
    for (int src = 1; src < WK; src++) {
      __syncthreads();
      if (wk == src) {
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
          for (int mo = 0; mo < MTO; mo++)
#pragma unroll
            for (int ni = 0; ni < NTI; ni++) red[((t * MTO + mo) * NTI + ni) * 64 + lane] = acc[t][mo][ni];
      }
      __syncthreads();
      if (wk == 0) {
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
          for (int mo = 0; mo < MTO; mo++)
#pragma unroll
            for (int ni = 0; ni < NTI; ni++) acc[t][mo][ni] += red[((t * MTO + mo) * NTI + ni) * 64 + lane];
      }
    }
  }
  if (wk == 0) {
    // lane: 4 consecutive co (rows 4*(lane>>4)+r) of one ci (col lane&15); slab [split][tap][ci][co]
    float* sl = slab + (int64_t)bx * 9 * Cin * Cout;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int mo = 0; mo < MTO; mo++)
#pragma unroll
        for (int ni = 0; ni < NTI; ni++) {
          const int co = co0 + (wco * MTO + mo) * 16 + 4 * (lane >> 4);
          const int ci = ci0 + (wci * NTI + ni) * 16 + (lane & 15);
          if (co < Cout && ci < Cin)
            *reinterpret_cast<f32x4*>(sl + ((int64_t)t * Cin + ci) * Cout + co) = acc[t][mo][ni];
        }
  }
#if MIS_TIMELINE
  TL_STAMP(30)
  if (tl_slot >= 0) mis_tl_buf[tl_slot][1023] = (unsigned long long)tl_n;
#endif
}

// dw[co][ci][tap] = sum_s slab[s][tap][ci][co].  RL row lanes x (1024/RL) columns per workgroup:
// many lanes per column when there are many splits and few outputs, one lane per column
// (pure streaming) when the weight tensor is large and the splits few.  Fixed order.
// One workgroup's share of that reduction: columns [blk*COLS, (blk+1)*COLS) of the 9*Cin*Cout outputs.
// `smem` holds RL x (COLS+1) doubles (RL > 1).
template <int RL>
__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ slab, int nsplit, int Cin, int Cout,
                                                   float* __restrict__ dw, int blk, double* smem) {
  constexpr int COLS = 1024 / RL;
  const int n = 9 * Cin * Cout;
  const int c = threadIdx.x % COLS, r = threadIdx.x / COLS;
  const int j = blk * COLS + c;
  double s = 0.0;
  if (j < n) {
    int i = r;
    for (; i + 3 * RL < nsplit; i += 4 * RL) {
      const float a0 = slab[(int64_t)i * n + j], a1 = slab[(int64_t)(i + RL) * n + j];
      const float a2 = slab[(int64_t)(i + 2 * RL) * n + j], a3 = slab[(int64_t)(i + 3 * RL) * n + j];
      s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    }
    for (; i < nsplit; i += RL) s += (double)slab[(int64_t)i * n + j];
  }
  if constexpr (RL > 1) {
    smem[r * (COLS + 1) + c] = s;
    __syncthreads();
    if (r != 0) return;
    s = 0.0;
#pragma unroll
    for (int k = 0; k < RL; k++) s += smem[k * (COLS + 1) + c];
  }
  if (j < n) {
    const int co = j % Cout, ci = (j / Cout) % Cin, tap = j / (Cout * Cin);
    dw[((int64_t)co * Cin + ci) * 9 + tap] = (float)s;
  }
}

constexpr int WGRAD_REDUCE_SMEM = 32 * 33 > 4 * 257 ? 32 * 33 : 4 * 257;     // doubles: the largest of the forms below

template <int RL>
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, int Cin,
                                                            int Cout, float* __restrict__ dw, int nblk_w,
                                                            BiasRows bias) {
  __shared__ double sm[WGRAD_REDUCE_SMEM];
  if ((int)blockIdx.x >= nblk_w) {        // rider: the conv-bias gradient from its partial rows
    reduce_rows_block(bias.rows, bias.n_rows, Cout, Cout, bias.dbias, (int)blockIdx.x - nblk_w,
                      reinterpret_cast<double(*)[33]>(sm));
    return;
  }
  wgrad_reduce_block<RL>(slab, nsplit, Cin, Cout, dw, (int)blockIdx.x, sm);
}

// Rows per column and workgroups of one layer's reduction (the three forms above).
struct WgReduceShape {
  int rl, nblk_w, nblk_bias;
};
inline WgReduceShape wgrad_reduce_shape(int nsplit, int Cin, int Cout, bool with_bias) {
  const int nout = 9 * Cin * Cout;
  WgReduceShape r;
  r.rl = (nsplit <= 16 || nout >= (1 << 18)) ? 1 : (nsplit <= 128 ? 4 : 32);
  r.nblk_w = cdiv(nout, 1024 / r.rl);
  r.nblk_bias = with_bias ? cdiv(Cout, 32) : 0;
  return r;
}

// The reductions of SEVERAL layers in one launch: layer j owns workgroups [first[j], first[j+1]).
constexpr int WGRAD_MULTI_MAX = 32;
struct WgMultiArgs {
  mis_wgrad_reduce_job job[WGRAD_MULTI_MAX];
  int first[WGRAD_MULTI_MAX + 1];
  int rl[WGRAD_MULTI_MAX], nblk_w[WGRAD_MULTI_MAX];
  int n;
};
__global__ __launch_bounds__(1024) void wgrad_reduce_multi_kernel(const WgMultiArgs a) {
  __shared__ double sm[WGRAD_REDUCE_SMEM];
  int j = 0;
  while (j + 1 < a.n && (int)blockIdx.x >= a.first[j + 1]) j++;
  const mis_wgrad_reduce_job& jb = a.job[j];
  const int blk = (int)blockIdx.x - a.first[j];
  if (blk >= a.nblk_w[j]) {
    reduce_rows_block(jb.bias_rows, jb.n_bias_rows, jb.Cout, jb.Cout, jb.dbias, blk - a.nblk_w[j],
                      reinterpret_cast<double(*)[33]>(sm));
    return;
  }
  if (a.rl[j] == 1) wgrad_reduce_block<1>(jb.slabs, jb.n_slabs, jb.Cin, jb.Cout, jb.dw, blk, sm);
  else if (a.rl[j] == 4) wgrad_reduce_block<4>(jb.slabs, jb.n_slabs, jb.Cin, jb.Cout, jb.dw, blk, sm);
  else wgrad_reduce_block<32>(jb.slabs, jb.n_slabs, jb.Cin, jb.Cout, jb.dw, blk, sm);
}

struct WgPlan {
  int cfg;          // 0: 1x1 tiles, K-split waves; 1: 2x1 (co 32 x ci 16); 2: 1x2; 3: 2x2 K-split; 4: 64x64 block
  int CTO, CTI, n_co_blocks, n_ci_blocks, tiles_x, tiles_y, n_tiles, nsplit;
  size_t lds;
  bool ws;          // wave-specialised form (cfg 3 and 4, bf16, no dilation): producer waves + two LDS stages
};

template <typename T>
bool plan_wgrad(int B, int H, int W, int Cin, int Cout, WgPlan& p, int Ca = -1, int dil = 1) {
  constexpr int G = Mf<T>::G;
  if (Cin % G != 0 || Cout % G != 0) return false;
  if ((int64_t)H * W * (Cin > Cout ? Cin : Cout) * (int64_t)sizeof(T) >= ((int64_t)1 << 31) - (1 << 22)) return false;
  if (Cout >= 64 && Cin >= 64) { p.cfg = 4; p.CTO = 64; p.CTI = 64; }
  else if (Cout > 16 && Cin > 16) { p.cfg = 3; p.CTO = 32; p.CTI = 32; }
  else if (Cout > 16) { p.cfg = 1; p.CTO = 32; p.CTI = 16; }
  else if (Cin > 16) { p.cfg = 2; p.CTO = 16; p.CTI = 32; }
  else { p.cfg = 0; p.CTO = 16; p.CTI = 16; }
  p.ws = sizeof(T) == 2 && dil == 1 && p.cfg >= 3;
  const int TH = wg_tile_h<T>(p.ws, p.CTO, p.CTI);
  if (Ca >= 0 && Ca != Cin && (Ca <= 0 || Ca % p.CTI != 0)) return false;   // channel tiles must not straddle the two inputs
  p.n_co_blocks = (Cout + p.CTO - 1) / p.CTO;
  p.n_ci_blocks = (Cin + p.CTI - 1) / p.CTI;
  p.tiles_x = (W + WG_TW - 1) / WG_TW;
  p.tiles_y = (H + TH - 1) / TH;
  p.n_tiles = B * p.tiles_x * p.tiles_y;
  // K-splits: enough workgroups to fill the chip, but the slab traffic (nsplit * dW bytes, written
  // and read once) must stay below the activation traffic of the layer
  const int nout = p.n_co_blocks * p.n_ci_blocks;
  const double act_bytes = (double)B * H * W * (Cin + Cout) * sizeof(T);
  const double dw_bytes = 36.0 * Cin * Cout;
  // one round of resident workgroups on 256 CUs: the four-wave variants fit three per CU, the
  // 32x32 K-split variant two, the 64x64 block (eight waves) one.
  const int target = (p.cfg == 4 || p.ws) ? 256 : p.cfg == 3 ? 512 : 768;
  int ns = target / nout;
  const int ns_slab = (int)(act_bytes / dw_bytes);
  if (ns > ns_slab) ns = ns_slab;
  const int ns_fill = (256 + nout - 1) / nout;          // at least one workgroup per CU
  if (ns < ns_fill) ns = ns_fill;
  if (g_wgrad_workgroups > 0) ns = g_wgrad_workgroups / nout;      // mis_set_conv_launch
  if (ns > p.n_tiles) ns = p.n_tiles;
  if (ns > 1024) ns = 1024;
  if (ns < 1) ns = 1;
  if (nout > 1 && ns >= 8) ns &= ~7;          // whole groups of 8: the kernel's XCD-aware id mapping
  p.nsplit = ns;
  const size_t stage = (size_t)(TH + 2 * dil) * (WG_TW + 2 * dil) * padded_row_bytes(p.CTI * (int)sizeof(T)) + 16 +
                       (size_t)TH * WG_TW * padded_row_bytes(p.CTO * (int)sizeof(T)) + 16;
  const size_t tiles = (p.ws ? 2 : 1) * stage + 2 * (size_t)p.CTI * 4;
  // K-split reduction scratch: one 9 x (tiles per wave) x 1 KiB region per (co,ci) wave group
  // (none in the wave-specialised 64x64 block: its waves do not split K)
  const size_t redb = (p.ws && p.cfg == 4) ? 0 : (size_t)9 * (p.CTO / 16) * (p.CTI / 16) * 64 * 16;
  p.lds = tiles > redb ? tiles : redb;
  if (p.lds > 160 * 1024) return false;
  return true;
}

// n_slabs_out: not null = write the slabs only and report their number (the caller reduces them later)
template <typename T>
int run_wgrad(const ConvIn& in, const void* dz, int B, int H, int W, int Cout, float* dw, const BiasRows& bias,
              void* workspace, hipStream_t st, int dil, int* n_slabs_out = nullptr) {
  WgPlan p;
  const int Cin = in.channels();
  if (dil < 0 || dil > 3) {
    mis_set_error("conv3x3_wgrad_mfma: dilation %d not supported (0..3)", dil);
    return MIS_ERR_UNSUPPORTED;
  }
  if (!plan_wgrad<T>(B, H, W, Cin, Cout, p, in.Ca, dil)) {
    mis_set_error("conv3x3_wgrad_mfma: unsupported shape Cin=%d Cout=%d", Cin, Cout);
    return MIS_ERR_UNSUPPORTED;
  }
  float* slab = (float*)workspace;
  const int nout_blocks = p.n_co_blocks * p.n_ci_blocks;
  dim3 grid(p.nsplit * nout_blocks);
#define WG_LAUNCH(MTO, NTI, WCO, WCI, WK)                                                                  \
  {                                                                                                        \
    auto kfn = dil == 2 ? conv3x3_wgrad_mfma_kernel<T, MTO, NTI, WCO, WCI, WK, 2>                          \
               : dil == 3 ? conv3x3_wgrad_mfma_kernel<T, MTO, NTI, WCO, WCI, WK, 3>                        \
               : dil == 0 ? conv3x3_wgrad_mfma_kernel<T, MTO, NTI, WCO, WCI, WK, 0>                        \
                          : conv3x3_wgrad_mfma_kernel<T, MTO, NTI, WCO, WCI, WK, 1>;                       \
    if (p.lds > 48 * 1024)                                                                                 \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds); \
    kfn<<<grid, 64 * WCO * WCI * WK, p.lds, st>>>((const T*)in.xa, (const T*)in.xb, in.Ca, (const T*)dz, slab, B, \
                                                  H, W, Cin, Cout, p.n_ci_blocks, p.tiles_x, p.tiles_y,    \
                                                  p.n_tiles, in.affa, in.affb, p.nsplit, nout_blocks);     \
  }
#define WG_LAUNCH_WS(MTO, NTI, WCO, WCI, WK)                                                               \
  {                                                                                                        \
    auto kfn = conv3x3_wgrad_mfma_kernel<T, MTO, NTI, WCO, WCI, WK, 1, true>;                              \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);   \
    kfn<<<grid, 64 * WCO * WCI * WK + WG_PRODUCERS, p.lds, st>>>(                                          \
        (const T*)in.xa, (const T*)in.xb, in.Ca, (const T*)dz, slab, B, H, W, Cin, Cout, p.n_ci_blocks,    \
        p.tiles_x, p.tiles_y, p.n_tiles, in.affa, in.affb, p.nsplit, nout_blocks);                         \
  }
  if constexpr (sizeof(T) == 2) {
    if (p.ws) {
      if (p.cfg == 3) WG_LAUNCH_WS(2, 1, 1, 2, 2)
      else WG_LAUNCH_WS(2, 2, 2, 2, 1)
    }
  }
  if (!p.ws) {
    switch (p.cfg) {
      case 0: WG_LAUNCH(1, 1, 1, 1, 4); break;
      case 1: WG_LAUNCH(2, 1, 1, 1, 4); break;
      case 2: WG_LAUNCH(1, 2, 1, 1, 4); break;
      case 3: WG_LAUNCH(2, 1, 1, 2, 2); break;
      default: WG_LAUNCH(2, 2, 2, 2, 2); break;
    }
  }
#undef WG_LAUNCH
#undef WG_LAUNCH_WS
  MIS_CHECK_LAUNCH("conv3x3_wgrad_mfma_kernel");
  if (n_slabs_out) {
    *n_slabs_out = p.nsplit;
    return MIS_OK;
  }
  const WgReduceShape r = wgrad_reduce_shape(p.nsplit, Cin, Cout, bias.dbias != nullptr);
  if (r.rl == 1)
    wgrad_reduce_kernel<1><<<r.nblk_w + r.nblk_bias, 1024, 0, st>>>(slab, p.nsplit, Cin, Cout, dw, r.nblk_w, bias);
  else if (r.rl == 4)
    wgrad_reduce_kernel<4><<<r.nblk_w + r.nblk_bias, 1024, 0, st>>>(slab, p.nsplit, Cin, Cout, dw, r.nblk_w, bias);
  else
    wgrad_reduce_kernel<32><<<r.nblk_w + r.nblk_bias, 1024, 0, st>>>(slab, p.nsplit, Cin, Cout, dw, r.nblk_w, bias);
  MIS_CHECK_LAUNCH("wgrad_reduce_kernel");
  return MIS_OK;
}

}  // namespace

bool conv3x3_wgrad_mfma_supported(int dtype, int B, int H, int W, int Cin, int Cout) {
  WgPlan p;
  return dtype == MIS_BF16 ? plan_wgrad<bf16>(B, H, W, Cin, Cout, p) : plan_wgrad<float>(B, H, W, Cin, Cout, p);
}

// Slab bytes of the largest plan any entry point can run for this shape: the dilated forms are never
// wave-specialised and may split K further than the plain one.
size_t conv3x3_wgrad_mfma_workspace_bytes(int dtype, int B, int H, int W, int Cin, int Cout) {
  int nsplit = 0;
  for (int dil = 0; dil <= 3; dil++) {
    WgPlan p;
    const bool ok = dtype == MIS_BF16 ? plan_wgrad<bf16>(B, H, W, Cin, Cout, p, -1, dil)
                                      : plan_wgrad<float>(B, H, W, Cin, Cout, p, -1, dil);
    if (ok && p.nsplit > nsplit) nsplit = p.nsplit;
  }
  return nsplit ? align256((size_t)nsplit * 9 * Cin * Cout * sizeof(float)) : 0;
}

void conv3x3_mfma_set_launch(int fwd_workgroups, int wgrad_workgroups) {
  g_fwd_workgroups = fwd_workgroups;
  g_wgrad_workgroups = wgrad_workgroups;
}

bool conv3x3_wgrad_mfma_cat_supported(int dtype, int B, int H, int W, int Ca, int Cb, int Cout) {
  WgPlan p;
  return dtype == MIS_BF16 ? plan_wgrad<bf16>(B, H, W, Ca + Cb, Cout, p, Ca)
                           : plan_wgrad<float>(B, H, W, Ca + Cb, Cout, p, Ca);
}

int conv3x3_wgrad_mfma(const ConvIn& in, const void* dz, int dtype, int B, int H, int W, int Cout, float* dw,
                       const BiasRows& bias, void* workspace, hipStream_t st, int dil) {
  if (dtype == MIS_BF16) return run_wgrad<bf16>(in, dz, B, H, W, Cout, dw, bias, workspace, st, dil);
  return run_wgrad<float>(in, dz, B, H, W, Cout, dw, bias, workspace, st, dil);
}

int conv3x3_wgrad_mfma_slabs(const ConvIn& in, const void* dz, int dtype, int B, int H, int W, int Cout, void* slabs,
                             int* n_slabs, hipStream_t st, int dil) {
  const BiasRows none{nullptr, 0, nullptr};
  if (dtype == MIS_BF16) return run_wgrad<bf16>(in, dz, B, H, W, Cout, nullptr, none, slabs, st, dil, n_slabs);
  return run_wgrad<float>(in, dz, B, H, W, Cout, nullptr, none, slabs, st, dil, n_slabs);
}

int conv3x3_wgrad_reduce_multi(const mis_wgrad_reduce_job* jobs, int n_jobs, hipStream_t st) {
  for (int j0 = 0; j0 < n_jobs; j0 += WGRAD_MULTI_MAX) {
    WgMultiArgs a;
    a.n = n_jobs - j0 < WGRAD_MULTI_MAX ? n_jobs - j0 : WGRAD_MULTI_MAX;
    int total = 0;
    for (int j = 0; j < a.n; j++) {
      const mis_wgrad_reduce_job& jb = jobs[j0 + j];
      const WgReduceShape r = wgrad_reduce_shape(jb.n_slabs, jb.Cin, jb.Cout, jb.dbias != nullptr);
      a.job[j] = jb;
      a.first[j] = total;
      a.rl[j] = r.rl;
      a.nblk_w[j] = r.nblk_w;
      total += r.nblk_w + r.nblk_bias;
    }
    a.first[a.n] = total;
    wgrad_reduce_multi_kernel<<<total, 1024, 0, st>>>(a);
    MIS_CHECK_LAUNCH("wgrad_reduce_multi_kernel");
  }
  return MIS_OK;
}

#if MIS_TIMELINE
extern "C" int mis_debug_timeline(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mis_tl_buf), sizeof(mis_tl_buf));
}
#endif
