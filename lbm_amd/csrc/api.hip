// C ABI of liblbm_hip.so (include/lbm_hip.h) and the per-frame step orchestration.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/lbm_hip.h"
#include "attn.hpp"
#include "common.hpp"
#include "encoder.hpp"
#include "final.hpp"
#include "gemm.hpp"
#include "msda.hpp"
#include "pyramid.hpp"
#include "rows.hpp"

namespace lbm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return -1;
  }
  return 0;
}

const char* intern_name(const char* s) {
  static std::mutex mu;
  static std::set<std::string> names;
  std::lock_guard<std::mutex> lk(mu);
  return names.insert(s).first->c_str();
}

// ---- per-launch timing ---------------------------------------------------------------------------
struct ProfRec {
  const char* name;
  hipEvent_t e0, e1;
  double flops, bytes;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static const size_t kProfMax = 1 << 16;

ProfScope::ProfScope(const char* name, hipStream_t s, double flops, double bytes) : slot(-1), stream(s) {
  if (!g_prof_on || g_prof.size() >= kProfMax) return;
  ProfRec r{name, nullptr, nullptr, flops, bytes};
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
  (void)hipEventRecord(r.e0, s);
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof[slot].e1, stream);
}

// ---- workspace layout of the step --------------------------------------------------------------
static constexpr unsigned long long Step_kSlabRows = 16384;
static constexpr unsigned long long Step_kFfnRowsMax = 16384;   // most rows for the many-rows fused feed-forward (its 8 slabs: 128 MB)


struct StepLayout {
  long long R, Q, L, P0, kmax;
  int n_vmaps;
  unsigned long long preq_bytes = 0;
  struct Item {
    const char* name;
    unsigned long long off, bytes;
  };
  Item items[48];
  int n = 0;
  unsigned long long total = 0;
  unsigned long long add(const char* name, unsigned long long bytes) {
    const unsigned long long off = total;
    items[n++] = Item{name, off, bytes};
    total += (bytes + 255ull) & ~255ull;
    return off;
  }
  long long find(const char* name) const {
    for (int i = 0; i < n; ++i)
      if (strcmp(items[i].name, name) == 0) return (long long)items[i].off;
    return -1;
  }
};

static int build_layout(const lbm_hip_step_desc* d, StepLayout& lay) {
  LBM_REQUIRE(d != nullptr, "step: null descriptor");
  LBM_REQUIRE(d->B >= 1 && d->N >= 0 && d->M >= 1 && d->M <= 4096, "step: bad B/N/M (%d, %d, %d)", d->B, d->N, d->M);
  LBM_REQUIRE(d->n_layers >= 1 && d->n_layers <= LBM_HIP_MAX_LAYERS, "step: n_layers=%d out of range", d->n_layers);
  LBM_REQUIRE(d->precision >= 0 && d->precision <= 3, "step: precision must be 0 (fp32 MFMA), 1 (split f16 MFMA), 2 (plain f16 MFMA) or 3 (f16 storage)");
  LBM_REQUIRE(d->stride == 4 && d->net_h % 32 == 0 && d->net_w % 32 == 0 && d->net_h > 0 && d->net_w > 0,
              "step: network input %dx%d (stride %d) must be a multiple of 32 with stride 4", d->net_h, d->net_w, d->stride);
  int kmax = 1;
  for (int i = 0; i < d->n_layers; ++i) {
    LBM_REQUIRE(d->topk[i] >= 1 && d->topk[i] <= 16, "step: topk[%d]=%d out of range", i, d->topk[i]);
    if (d->topk[i] > kmax) kmax = d->topk[i];
  }
  LBM_REQUIRE(d->topk[d->n_layers - 1] == 1, "step: the last layer must select exactly one region (transformer.py:17)");
  const int h = d->net_h / d->stride, w = d->net_w / d->stride;
  lay.P0 = (long long)h * w;
  lay.L = lay.P0 + lay.P0 / 4 + lay.P0 / 16 + lay.P0 / 64;
  lay.R = (long long)d->B * d->N;
  lay.kmax = kmax;
  lay.Q = lay.R * kmax;
  lay.n_vmaps = d->n_layers + 3;
  const unsigned long long f = sizeof(float);
  lay.add("value", (unsigned long long)d->B * lay.L * lay.n_vmaps * LBM_C * (d->precision == 3 ? 2 : f));      // (precision 3: f16 value maps)
  lay.add("inv_norm", (unsigned long long)d->B * lay.L * f);
  lay.add("corr_maps", (unsigned long long)d->n_layers * d->B * lay.P0 * LBM_C * f);
  lay.add("corr", (unsigned long long)lay.R * lay.P0 * f);
  lay.add("U", (unsigned long long)lay.R * 1024 * f);
  lay.add("mbar", (unsigned long long)lay.R * 1024 * f);
  lay.add("hid", (unsigned long long)lay.Q * 1024 * f);
  lay.add("ta", (unsigned long long)lay.R * LBM_C * f);
  lay.add("tb", (unsigned long long)lay.R * LBM_C * f);
  lay.add("tc", (unsigned long long)lay.R * LBM_C * f);
  lay.add("x0", (unsigned long long)lay.R * LBM_C * f);
  lay.add("x1", (unsigned long long)lay.R * LBM_C * f);
  lay.add("qn", (unsigned long long)lay.R * LBM_C * f);
  lay.add("valid", (unsigned long long)lay.R);
  lay.add("idx", (unsigned long long)lay.Q * sizeof(int));
  lay.add("ref", (unsigned long long)lay.Q * 2 * f);
  lay.add("offlog", (unsigned long long)lay.R * (48 * 16 + 1024) * f);      // deformable offsets/weights (+ merged column blocks)
  lay.add("samp", (unsigned long long)lay.Q * LBM_C * f);
  lay.add("la", (unsigned long long)lay.Q * LBM_C * f);
  lay.add("lb", (unsigned long long)lay.Q * LBM_C * f);
  lay.add("pred", (unsigned long long)lay.R * 2 * f);
  lay.add("coll_tok", (unsigned long long)lay.R * LBM_C * f);
  lay.add("preq", (unsigned long long)lay.Q * LBM_C * f);                   // pre-LayerNorm rows
  // K-split partial products, one region per sub-run; an undivided run may use all of it for the eight slabs of the many-rows
  // feed-forward (ffn_rows, up to Step_kFfnRowsMax rows)
  {
    const unsigned long long q = (unsigned long long)(lay.Q < (long long)Step_kFfnRowsMax ? lay.Q : (long long)Step_kFfnRowsMax);
    const unsigned long long rows = 8ull * q > 4ull * Step_kSlabRows ? 8ull * q : 4ull * Step_kSlabRows;
    lay.add("slab", rows * LBM_C * f);
  }
  lay.add("hidR", (unsigned long long)lay.R * LBM_C * f);
  if (d->precision >= 1) {   // split twins (two f16 planes = the same number of bytes as the fp32 tensor)
    if (d->tokens_split == nullptr) lay.add("tok_s", (unsigned long long)d->B * lay.L * LBM_C * f);
    lay.add("q_s", (unsigned long long)lay.R * LBM_C * f);
    lay.add("x0_s", (unsigned long long)lay.R * LBM_C * f);
    lay.add("x1_s", (unsigned long long)lay.R * LBM_C * f);
    lay.add("ta_s", (unsigned long long)lay.R * LBM_C * f);
    lay.add("tb_s", (unsigned long long)lay.R * LBM_C * f);
    lay.add("tc_s", (unsigned long long)lay.R * LBM_C * f);
    lay.add("la_s", (unsigned long long)lay.Q * LBM_C * f);
    lay.add("lb_s", (unsigned long long)lay.Q * LBM_C * f);
  }
  return 0;
}

// A tensor view used by the step: fp32 rows and/or the same values as split f16 planes (hi at `s`, lo at s + plane).
static int g_rs_topk = 1;           // tuning (debug key 13): 0 = the correlation product never takes the row-streaming kernel with per-point lists
static int g_rs_dbg = 0;            // tuning (debug key 14): ablation bits handed to gemm_rs_topk
static int g_fuse_kmax_small = 1;   // tuning (debug key 6): largest k fused into the correlation product at < 4096 rows
static int g_ffn_sites = 63;        // tuning (debug key 8): 1 attention blocks, 2 lattice blocks, 4 head blocks may fuse
extern int g_ffn_rows_dbg;          // ffn.hip (debug key 16)
extern int g_conv64_dbg;            // conv_rs.hip (debug key 18)
extern int g_stem_dbg;              // stem_rs.hip (debug key 20)
extern int g_msda_xcd;              // msda.hip (debug key 21)
extern int g_msda_window;           // msda.hip (debug key 22)
extern int g_msda_dbg;              // msda.hip (debug key 23)
extern int g_tail_order;            // pyramid.hip (debug key 24)
extern int g_rs_ring;               // gemm_rs.hip (debug key 25)
extern int g_conv_patch_dbg;        // conv_patch.hip (debug key 26, bits 0-7)
extern int g_conv_patch_tile;       // conv_patch.hip (debug key 26, bits 8..)
static int g_head_pair = 1;         // tuning (debug key 17): 0 = offset and visibility heads as two separate chains
static int g_ffn_rows_mode = 0;     // tuning (debug key 15): 0 = many-rows fused feed-forward from g_ffn_fused_rows + 1 rows on, 1 = never, 2 = for any row count
static int g_ffn_fused_rows = 2048; // tuning (debug key 7): most rows for which the feed-forward runs as one fused launch (0: never)

// every knob lbm_hip_debug_set has been given, so that a measurement can state what was active (lbm_hip_debug_active)
static std::mutex g_knob_mu;
static std::map<int, int> g_knobs;
static int knob_default(int key) {
  switch (key) {
    case 6: case 13: case 17: case 21: case 24: return 1;
    case 22: case 25: return 3;
    case 7: return 2048;
    case 8: return 63;
    default: return 0;
  }
}

struct TV {
  float* f = nullptr;
  void* s = nullptr;
  long long plane = 0;
};

struct Step {
  const lbm_hip_step_desc* d;
  StepLayout lay;
  hipStream_t s;
  char* ws;
  int h, w;
  bool split;      // precision 1 / 2: dense contractions on the f16 matrix cores with split operands
  int planes = 2;  // 2: hi and lo planes multiplied (fp32-grade); 1: hi planes only (precision 2 / 3, reduced)
  int cplanes = 2; // the same for the correlation product alone: precision 3 keeps it fp32-grade so the arg-max is the fp32 one
  bool h16 = false; // precision 3: memories (caller's ring buffers) and the six value maps are f16 in HBM
  // Points never interact, so the per-point phase can run on any partition of the rows: this run covers
  // rows [lo, lo + R) of Rfull; `sub` of `nsub` names its private K-split slab region.
  long long Rfull = 0;
  int lo = 0, R = 0, sub = 0, nsub = 1;
  static constexpr long long kSlabRows = 16384;  // per-sub-run slab capacity (rows of 256 floats): K-split / feed-forward partial outputs
  long long slab_rows_total() const {               // the whole "slab" buffer (an undivided run may use all of it)
    const long long q = lay.Q < (long long)Step_kFfnRowsMax ? lay.Q : (long long)Step_kFfnRowsMax;
    return 8 * q > 4 * kSlabRows ? 8 * q : 4 * kSlabRows;
  }

  template <typename T>
  T* buf(const char* name) {
    return reinterpret_cast<T*>(ws + lay.find(name));
  }

  const float* W(int idx) const { return d->weights[idx]; }
  // distance (f16 elements) from the hi plane of matrix idx to its lo plane; unused when only hi planes are multiplied
  long long lo_plane(int idx) const {
    return planes == 2 ? ((const char*)d->weights_lo[idx] - (const char*)d->weights[idx]) / 2 : 0;
  }
  // rows-of-points buffers with capacity `cap` rows of width `wd` per point; this run owns the byte range of its
  // own points (so concurrent runs never touch each other's scratch, whatever `mult` <= cap each is using right
  // now) and lays `mult` rows per point out inside it.  fname: fp32 buffer; sname: split twin, hi plane at the start
  // of the range, lo plane R * mult * wd elements further.
  TV view(const char* fname, const char* sname, int cap, int mult, int wd) {
    TV t;
    const long long off = (long long)lo * cap * wd;      // in 4-byte units
    if (fname) t.f = buf<float>(fname) + off;
    if (split && sname) {
      t.s = buf<char>(sname) + off * 4;
      t.plane = (long long)R * mult * wd;
    }
    return t;
  }
  // buffers that hold ONE representation: fp32 on the exact path, split planes on the split path (same bytes)
  TV view1(const char* name, int cap, int mult, int wd) {
    return split ? view(nullptr, name, cap, mult, wd) : view(name, nullptr, cap, mult, wd);
  }

  // Y = act(X · W[wi]ᵀ + W[bi]) + res_scale · res[(m / res_div)]   (Y.f fp32 and / or Y.s split, as the view asks)
  int linear(const TV& X, int ldx, int wi, int bi, const TV& Y, int ldy, int M, int N, int K, int act = ACT_NONE,
             const float* res = nullptr, int res_div = 1, float res_scale = 1.f) {
    if (!split) {
      GemmArgs a;
      a.X = X.f; a.ldx = ldx; a.W = W(wi); a.ldw = K; a.bias = bi >= 0 ? W(bi) : nullptr; a.Y = Y.f; a.ldy = ldy;
      a.M = M; a.N = N; a.K = K; a.act = act; a.res = res; a.ldres = LBM_C; a.res_div = res_div; a.res_scale = res_scale;
      return gemm(a, s);
    }
    GemmSSArgs a;
    a.planes = planes;
    a.A = X.s; a.a_plane = X.plane; a.lda = ldx;
    a.B = d->weights[wi]; a.b_plane = lo_plane(wi);
    a.ldb = K; a.bias = bi >= 0 ? W(bi) : nullptr;
    a.Y = Y.f; a.ldy = ldy; a.Ys = Y.s; a.ys_plane = Y.plane; a.ldys = ldy;
    a.M = M; a.N = N; a.K = K; a.act = act; a.res = res; a.ldres = LBM_C; a.res_div = res_div; a.res_scale = res_scale;
    return gemm_ss(a, s);
  }

  // Y = LayerNorm(X · W[wi]ᵀ + W[bi] + res_scale · res[m / res_div]) (N = 256), rows with keep == 0 replaced by fallback.
  // The bias / residual / K-split reduction happen in the LayerNorm kernel: with few rows the product is split
  // along K over gridDim.z (partial slabs), which is what keeps all 256 CUs busy at N = 1024 points.
  // `mult`: rows per point of this product (its scratch is the row range of this run inside "preq").
  int linear_ln(const TV& X, int ldx, int wi, int bi, int mult, int M, int K, const float* res, int res_div, float res_scale,
                int gi, int bei, const TV& Y, const uint8_t* keep = nullptr, const float* fallback = nullptr, const TV* Yn = nullptr) {
    (void)mult;
    float* pre = buf<float>("preq") + (long long)lo * lay.kmax * LBM_C;
    RowsIn in;
    in.bias = W(bi); in.res = res; in.res_div = res_div; in.res_scale = res_scale;
    if (!split) {
      GemmArgs a;
      a.X = X.f; a.ldx = ldx; a.W = W(wi); a.ldw = K; a.Y = pre; a.ldy = LBM_C; a.M = M; a.N = LBM_C; a.K = K;
      if (gemm(a, s)) return -1;
    } else {
      int nsplit = 1;
      const long long tiles = (long long)((M + 63) / 64) * (LBM_C / 64);
      while (nsplit < 8 && tiles * nsplit < 192 && K / (nsplit * 2) >= 256 && (K / (nsplit * 2)) % 32 == 0 &&
             (long long)nsplit * 2 * M <= kSlabRows)
        nsplit *= 2;
      if (nsplit > 1) pre = buf<float>("slab") + (long long)sub * kSlabRows * LBM_C;
      GemmSSArgs a;
      a.planes = planes;
      a.A = X.s; a.a_plane = X.plane; a.lda = ldx;
      a.B = d->weights[wi]; a.b_plane = lo_plane(wi); a.ldb = K;
      a.Y = pre; a.ldy = LBM_C; a.M = M; a.N = LBM_C; a.K = K / nsplit;
      a.batch = nsplit; a.sA = K / nsplit; a.sB = K / nsplit; a.sY = (long long)M * LBM_C;
      if (gemm_ss(a, s)) return -1;
      in.nslab = nsplit; in.slab_stride = (long long)M * LBM_C;
    }
    in.X = pre;
    RowsOut o;
    o.Y = Y.f; o.Ys = Y.s; o.ys_plane = Y.plane;
    if (Yn) {
      o.Yn = Yn->f; o.Yns = Yn->s; o.yns_plane = Yn->plane;
    }
    return layernorm_rows_in(in, W(gi), W(bei), M, keep, fallback, o, s);
  }

  // Y = LayerNorm(MLP(X) + res) with MLP = F2 . GELU(F1 . x + b1) + b2 (transformer.py:343-355 inside every block).
  // Few rows: one fused launch over the fragment-major weights leaves eight partial outputs for the LayerNorm kernel
  // (ffn.hip); otherwise two products, the second split along K.  `wb` is the block's table base, f1w.. its offsets.
  int ffn_ln(const TV& X, int wb, int f1w, int f1b, int f2w, int f2b, int f1fm, int f2fm, int g, int be, int M, const float* res,
             const TV& Y, const uint8_t* keep = nullptr, const float* fallback = nullptr, const TV* Yn = nullptr, const TV* hid = nullptr,
             int site = 1) {
    const bool fused = (g_ffn_sites & site) && split && M <= g_ffn_fused_rows && 8ll * M <= kSlabRows && W(wb + f1fm) != nullptr && W(wb + f2fm) != nullptr;
    // many rows: the same eight slabs from the kernel that keeps the weights in registers and streams the rows (ffn.hip)
    const bool rows_path = (g_ffn_sites & site) && split && planes == 2 && nsub == 1 && g_ffn_rows_mode != 1 &&
                           (g_ffn_rows_mode == 2 || M > g_ffn_fused_rows) && M <= (long long)Step_kFfnRowsMax && 8ll * M <= slab_rows_total() &&
                           W(wb + f1fm) != nullptr && W(wb + f2fm) != nullptr;
    if (!fused && !rows_path) {
      if (linear(X, LBM_C, wb + f1w, wb + f1b, *hid, 1024, M, 1024, LBM_C, ACT_GELU)) return -1;
      return linear_ln(*hid, 1024, wb + f2w, wb + f2b, 1, M, 1024, res, 1, 1.f, wb + g, wb + be, Y, keep, fallback, Yn);
    }
    FfnArgs f;
    f.A = X.s; f.a_plane = X.plane; f.lda = LBM_C; f.F1 = d->weights[wb + f1fm]; f.b1 = W(wb + f1b); f.F2 = d->weights[wb + f2fm];
    f.slabs = buf<float>("slab") + (long long)sub * kSlabRows * LBM_C; f.slab_stride = (long long)M * LBM_C;
    f.M = M; f.chunk = 128; f.planes = planes;
    if (rows_path ? ffn_rows(f, s) : ffn_fused(f, s)) return -1;
    RowsIn in;
    in.X = f.slabs; in.nslab = 8; in.slab_stride = f.slab_stride; in.bias = W(wb + f2b); in.res = res; in.res_div = 1; in.res_scale = 1.f;
    RowsOut o;
    o.Y = Y.f; o.Ys = Y.s; o.ys_plane = Y.plane;
    if (Yn) {
      o.Yn = Yn->f; o.Yns = Yn->s; o.yns_plane = Yn->plane;
    }
    return layernorm_rows_in(in, W(wb + g), W(wb + be), M, keep, fallback, o, s);
  }

  // memory / lattice-update attention block: x_in [R,256] -> x_out [R,256]
  // U_pre / ldu: the queries u_h, when an earlier merged product already holds them (a column block of pitch ldu)
  int attn_block(int wb, const TV& x_in, const TV& x_out, const float* seq, long long point_stride, int S, int ring_head,
                 bool masked, const TV* Yn, const TV* seq_tv = nullptr, const float* U_pre = nullptr, long long ldu = 1024,
                 float* U_store = nullptr) {
    const int km = (int)lay.kmax;
    TV U = view("U", nullptr, 1, 1, 1024);
    TV mbar = view1("mbar", 1, 1, 1024);
    TV hid = view1("hid", km, 1, 1024);
    TV y = view("tc", "tc_s", 1, 1, LBM_C);
    uint8_t* valid = masked ? buf<uint8_t>("valid") + lo : nullptr;
    if (!masked && S == 1 && seq_tv != nullptr) {
      // one key: its soft-max weight is exactly 1 in every head, so the read-out is the key itself and
      // Hm . [key; key; key; key] = (sum over heads of the Hm blocks) . key -- no query projection, no read-out, K = 256
      if (linear_ln(*seq_tv, LBM_C, wb + LBM_W_ATTN_HMSUM, wb + LBM_W_ATTN_HB, 1, R, LBM_C, x_in.f, 1, 1.f, wb + LBM_W_ATTN_LN1G,
                    wb + LBM_W_ATTN_LN1B, y))
        return -1;
      return ffn_ln(y, wb, LBM_W_ATTN_F1W, LBM_W_ATTN_F1B, LBM_W_ATTN_F2W, LBM_W_ATTN_F2B, LBM_W_ATTN_F1FM, LBM_W_ATTN_F2FM,
                    LBM_W_ATTN_LN2G, LBM_W_ATTN_LN2B, R, y.f, x_out, valid, x_in.f, Yn, &hid, 16);
    }
    if (U_store != nullptr) {      // the caller's cache of these queries: fill it now, read it from now on
      U.f = U_store;
      U_pre = nullptr;
    }
    if (U_pre == nullptr && linear(x_in, LBM_C, wb + LBM_W_ATTN_G, wb + LBM_W_ATTN_GB, U, 1024, R, 1024, LBM_C)) return -1;
    AttnArgs a;
    a.U = U_pre ? U_pre : U.f; a.ldu = U_pre ? ldu : 1024; a.seq = seq; a.point_stride = point_stride; a.slot_stride = LBM_C; a.S = S; a.ring_head = ring_head;
    a.seq_f16 = (masked && h16) ? 1 : 0;         // the two memories; the lattice tokens of the update attention stay fp32
    a.temb = masked ? W(wb + LBM_W_ATTN_TEMB) : nullptr;
    a.mask_a = masked ? d->vis_mask + (long long)lo * d->M : nullptr;
    a.mask_b = masked ? d->mem_mask + (long long)lo * d->M : nullptr;
    a.out = mbar.f; a.out_s = mbar.s; a.out_plane = mbar.plane; a.valid = valid; a.R = R;
    if (attn_readout(a, s)) return -1;
    if (linear_ln(mbar, 1024, wb + LBM_W_ATTN_HM, wb + LBM_W_ATTN_HB, 1, R, 1024, x_in.f, 1, 1.f, wb + LBM_W_ATTN_LN1G,
                  wb + LBM_W_ATTN_LN1B, y))
      return -1;
    return ffn_ln(y, wb, LBM_W_ATTN_F1W, LBM_W_ATTN_F1B, LBM_W_ATTN_F2W, LBM_W_ATTN_F2B, LBM_W_ATTN_F1FM, LBM_W_ATTN_F2FM,
                  LBM_W_ATTN_LN2G, LBM_W_ATTN_LN2B, R, y.f, x_out, valid, x_in.f, Yn, &hid, !masked ? 32 : Yn ? 8 : 1);
  }

  // deformable block: token x [R,256] (replicated k times) sampled at ref [R,k,2] -> out [R*k,256]
  // ol_pre / ld_ol: this block's offsets/weights, when an earlier merged product already holds them
  int msda_block(int wb, int vmap, int P, const TV& x, const float* ref, int k, const TV& out, const float* ol_pre = nullptr,
                 int ld_ol = 0) {
    const int Q = R * k;
    const int km = (int)lay.kmax;
    TV offlog = view("offlog", nullptr, 1, 1, 48 * 16 + 1024);
    TV samp = view1("samp", km, k, LBM_C);
    TV hid = view1("hid", km, k, 1024);
    TV la = view("la", "la_s", km, k, LBM_C);
    const int ncol = 48 * P;
    if (ol_pre == nullptr && linear(x, LBM_C, wb + LBM_W_MSDA_OLW, wb + LBM_W_MSDA_OLB, offlog, ncol, R, ncol, LBM_C)) return -1;
    MsdaArgs m;
    m.value = h16 ? reinterpret_cast<const float*>(buf<_Float16>("value") + (long long)vmap * LBM_C) : buf<float>("value") + (long long)vmap * LBM_C;
    m.value_f16 = h16 ? 1 : 0;
    m.ldv = lay.n_vmaps * LBM_C;
    m.v_batch_stride = lay.L * m.ldv;
    int st = 0;
    for (int l = 0; l < 4; ++l) {
      m.lvl_h[l] = h >> l; m.lvl_w[l] = w >> l; m.lvl_start[l] = st;
      st += (h >> l) * (w >> l);
    }
    // a sub-run only exists for B == 1, where its rows all belong to stream 0
    m.B = nsub > 1 ? 1 : d->B; m.N = nsub > 1 ? R : d->N; m.k = k; m.P = P;
    m.offlog = ol_pre ? ol_pre : offlog.f; m.ld_offlog = ol_pre ? ld_ol : ncol; m.logit_col = 32 * P;
    m.ref = ref; m.net_w = (float)d->net_w; m.net_h = (float)d->net_h;
    m.out = samp.f; m.out_s = samp.s; m.out_plane = samp.plane;
    if (msda_sample(m, s)) return -1;
    // 2*q + output_proj(.)  (msdeformattn.py:185 adds the identity, transformer.py:321 adds the query again)
    if (linear_ln(samp, LBM_C, wb + LBM_W_MSDA_OUTW, wb + LBM_W_MSDA_OUTB, k, Q, LBM_C, x.f, k, 2.f, wb + LBM_W_MSDA_LN1G,
                  wb + LBM_W_MSDA_LN1B, la))
      return -1;
    return ffn_ln(la, wb, LBM_W_MSDA_F1W, LBM_W_MSDA_F1B, LBM_W_MSDA_F2W, LBM_W_MSDA_F2B, LBM_W_MSDA_F1FM, LBM_W_MSDA_F2FM,
                  LBM_W_MSDA_LN2G, LBM_W_MSDA_LN2B, Q, la.f, out, nullptr, nullptr, nullptr, &hid, wb >= d->n_layers * LBM_W_LAYER_COUNT ? 4 : 2);
  }


  // Offset head and visibility head as ONE batch of two (transformer.py:103-149): both sample the same points at the same cell
  // centres with the same shapes and differ only in their value map and weights, so every launch of the pair -- sampling, output
  // projection, LayerNorm, feed-forward, LayerNorm, first layer of the read-out MLP -- covers 2 R rows with two parameter sets
  // (6 launches fewer per frame; identical arithmetic per row, so the results are bit-identical to the two separate chains).
  // Returns 1 when done (hidden layers of the two read-out MLPs in hid2[0:R] / hid2[R:2R]), 0 when not applicable, -1 on error.
  int heads_pair(int hb, const TV& x, const float* ref, const float* ol3, int c_off, int c_vis, int ld3, float** hid2_out) {
    const int wa = hb + LBM_W_HEAD_OFFSET, wbv = hb + LBM_W_HEAD_VIS;
    if (!g_head_pair || !split || planes != 2 || nsub != 1 || lo != 0 || ol3 == nullptr || lay.kmax < 2 || 16ll * R > kSlabRows) return 0;
    const int pairs[][2] = {{wa + LBM_W_MSDA_OUTW, wbv + LBM_W_MSDA_OUTW}, {hb + LBM_W_HEAD_OFFH_W0, hb + LBM_W_HEAD_VISH_W0}};
    for (const auto& pr : pairs) {
      const long long diff = (const char*)d->weights[pr[1]] - (const char*)d->weights[pr[0]];
      if (lo_plane(pr[0]) != lo_plane(pr[1]) || diff % 16 != 0) return 0;
    }
    if (W(wa + LBM_W_MSDA_F1FM) == nullptr || W(wa + LBM_W_MSDA_F2FM) == nullptr || W(wbv + LBM_W_MSDA_F1FM) == nullptr ||
        W(wbv + LBM_W_MSDA_F2FM) == nullptr || !(g_ffn_sites & 4))
      return 0;
    const long long RC = (long long)R * LBM_C;
    TV samp2, la2, out2;
    samp2.s = buf<char>("samp"); samp2.plane = 2 * RC;
    la2.f = buf<float>("la"); la2.s = buf<char>("la_s"); la2.plane = 2 * RC;
    out2.s = buf<char>("lb_s"); out2.plane = 2 * RC;
    float* pre = buf<float>("preq");
    float* hid2 = buf<float>("lb");
    // sampling: rows [0, R) from the offset head's value map and offsets, rows [R, 2R) from the visibility head's
    MsdaArgs m;
    m.value = buf<float>("value") + (long long)d->n_layers * LBM_C;
    m.set_value_off = 2ll * LBM_C;                       // value maps: n_layers = offset, + 1 = collision memory, + 2 = visibility
    m.ldv = lay.n_vmaps * LBM_C;
    m.v_batch_stride = lay.L * m.ldv;
    int st = 0;
    for (int l = 0; l < 4; ++l) {
      m.lvl_h[l] = h >> l; m.lvl_w[l] = w >> l; m.lvl_start[l] = st;
      st += (h >> l) * (w >> l);
    }
    m.B = d->B; m.N = d->N; m.k = 1; m.P = d->n_points;
    m.offlog = ol3 + c_off; m.ld_offlog = ld3; m.logit_col = 32 * d->n_points; m.set_offlog_off = c_vis - c_off;
    m.set_rows = R;
    m.ref = ref; m.net_w = (float)d->net_w; m.net_h = (float)d->net_h;
    m.out_s = samp2.s; m.out_plane = samp2.plane;
    if (msda_sample(m, s)) return -1;
    auto wdiff = [&](int ia, int ib) { return ((const char*)d->weights[ib] - (const char*)d->weights[ia]) / 2; };   // f16 elements
    // output projection of both (K = 256: no K-split, the batch dimension is free for the pair)
    {
      GemmSSArgs a;
      a.planes = 2;
      a.A = samp2.s; a.a_plane = samp2.plane; a.lda = LBM_C; a.sA = RC;
      a.B = d->weights[wa + LBM_W_MSDA_OUTW]; a.b_plane = lo_plane(wa + LBM_W_MSDA_OUTW); a.ldb = LBM_C;
      a.sB = wdiff(wa + LBM_W_MSDA_OUTW, wbv + LBM_W_MSDA_OUTW);
      a.Y = pre; a.ldy = LBM_C; a.sY = RC; a.M = R; a.N = LBM_C; a.K = LBM_C; a.batch = 2;
      if (gemm_ss(a, s)) return -1;
    }
    {   // 2 q + output_proj(.) -> LayerNorm  (msdeformattn.py:185, transformer.py:321)
      RowsIn in;
      in.X = pre; in.bias = W(wa + LBM_W_MSDA_OUTB); in.bias2 = W(wbv + LBM_W_MSDA_OUTB);
      in.res = x.f; in.res2 = x.f; in.res_div = 1; in.res_scale = 2.f; in.set_rows = R;
      in.gamma2 = W(wbv + LBM_W_MSDA_LN1G); in.beta2 = W(wbv + LBM_W_MSDA_LN1B);
      RowsOut o;
      o.Y = la2.f; o.Ys = la2.s; o.ys_plane = la2.plane;
      if (layernorm_rows_in(in, W(wa + LBM_W_MSDA_LN1G), W(wa + LBM_W_MSDA_LN1B), 2 * R, nullptr, nullptr, o, s)) return -1;
    }
    {   // feed-forward + LayerNorm
      FfnArgs f;
      f.A = la2.s; f.a_plane = la2.plane; f.lda = LBM_C;
      f.F1 = d->weights[wa + LBM_W_MSDA_F1FM]; f.b1 = W(wa + LBM_W_MSDA_F1B); f.F2 = d->weights[wa + LBM_W_MSDA_F2FM];
      f.F1b = d->weights[wbv + LBM_W_MSDA_F1FM]; f.b1b = W(wbv + LBM_W_MSDA_F1B); f.F2b = d->weights[wbv + LBM_W_MSDA_F2FM];
      f.set_rows = R;
      f.slabs = buf<float>("slab"); f.slab_stride = 2 * RC; f.M = 2 * R; f.chunk = 128; f.planes = 2;
      if (ffn_fused(f, s)) return -1;
      RowsIn in;
      in.X = f.slabs; in.nslab = 8; in.slab_stride = f.slab_stride;
      in.bias = W(wa + LBM_W_MSDA_F2B); in.bias2 = W(wbv + LBM_W_MSDA_F2B);
      in.res = la2.f; in.res2 = la2.f + RC; in.res_div = 1; in.res_scale = 1.f; in.set_rows = R;
      in.gamma2 = W(wbv + LBM_W_MSDA_LN2G); in.beta2 = W(wbv + LBM_W_MSDA_LN2B);
      RowsOut o;
      o.Ys = out2.s; o.ys_plane = out2.plane;
      if (layernorm_rows_in(in, W(wa + LBM_W_MSDA_LN2G), W(wa + LBM_W_MSDA_LN2B), 2 * R, nullptr, nullptr, o, s)) return -1;
    }
    {   // first layers of the two read-out MLPs (Linear + GELU)
      GemmSSArgs a;
      a.planes = 2;
      a.A = out2.s; a.a_plane = out2.plane; a.lda = LBM_C; a.sA = RC;
      a.B = d->weights[hb + LBM_W_HEAD_OFFH_W0]; a.b_plane = lo_plane(hb + LBM_W_HEAD_OFFH_W0); a.ldb = LBM_C;
      a.sB = wdiff(hb + LBM_W_HEAD_OFFH_W0, hb + LBM_W_HEAD_VISH_W0);
      a.bias = W(hb + LBM_W_HEAD_OFFH_B0); a.sBias = W(hb + LBM_W_HEAD_VISH_B0) - W(hb + LBM_W_HEAD_OFFH_B0);
      a.Y = hid2; a.ldy = LBM_C; a.sY = RC; a.M = R; a.N = LBM_C; a.K = LBM_C; a.batch = 2; a.act = ACT_GELU;
      if (gemm_ss(a, s)) return -1;
    }
    *hid2_out = hid2;
    return 1;
  }

  // ---- per-frame, point-independent work: split tokens, the six value maps, the correlation maps ----
  int run_frame() {
    const int hb = d->n_layers * LBM_W_LAYER_COUNT;
    TV value;
    if (h16) value.s = buf<char>("value");      // plain f16 rows (plane 0: no lo plane is written)
    else value.f = buf<float>("value");
    const int nv = lay.n_vmaps * LBM_C;
    const long long BL = (long long)d->B * lay.L;
    TV tok; tok.f = const_cast<float*>(d->tokens);
    if (split) {
      tok.plane = BL * LBM_C;
      if (d->tokens_split) {
        tok.s = const_cast<void*>(d->tokens_split);
      } else {
        tok.s = buf<char>("tok_s");
        if (split_planes(d->tokens, tok.s, (char*)tok.s + 2 * tok.plane, BL * LBM_C, s)) return -1;
      }
    }
    if (linear(tok, LBM_C, hb + LBM_W_HEAD_VPROJ_W, hb + LBM_W_HEAD_VPROJ_B, value, nv, (int)BL, nv, LBM_C)) return -1;
    // (row norms handed over by the encoder tail: the norm kernel is skipped)
    return corr_maps(d->tokens, d->inv_norm ? const_cast<float*>(d->inv_norm) : buf<float>("inv_norm"), buf<float>("corr_maps"), d->corr_mix,
                     d->n_layers, d->B, h, w, split, s, d->inv_norm != nullptr);
  }

  // ---- everything that depends on the points, for rows [lo, lo + R) ----
  int run_points() {
    if (R == 0) return 0;
    const int hb = d->n_layers * LBM_W_LAYER_COUNT;
    const long long RC = (long long)R * LBM_C;
    const int M = d->M;
    float* F = buf<float>("corr_maps");
    const float* query = d->query + (long long)lo * LBM_C;
    TV xq = view(nullptr, "q_s", 1, 1, LBM_C);
    xq.f = const_cast<float*>(query);
    if (split && d->query_split) {          // the caller keeps the twin: planes of the whole [Rfull, 256] query
      xq.s = (char*)const_cast<void*>(d->query_split) + (long long)lo * LBM_C * 2;
      xq.plane = Rfull * LBM_C;
    } else if (split && split_planes(query, xq.s, (char*)xq.s + 2 * xq.plane, RC, s)) {
      return -1;
    }
    TV x0 = view("x0", "x0_s", 1, 1, LBM_C), x1 = view("x1", "x1_s", 1, 1, LBM_C), ta = view("ta", "ta_s", 1, 1, LBM_C),
       tb = view("tb", "tb_s", 1, 1, LBM_C);
    TV qn = view1("qn", 1, 1, LBM_C);
    float* corr = buf<float>("corr") + (long long)lo * lay.P0;
    float* ref = buf<float>("ref") + (long long)lo * lay.kmax * 2;
    int* idx = buf<int>("idx") + (long long)lo * lay.kmax;
    // (precision 3: the ring buffers hold f16 elements -- same element offsets, half the bytes)
    const long long mem_off = (long long)lo * M * LBM_C;
    const float* coll_mem = h16 ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(d->coll_mem) + mem_off) : d->coll_mem + mem_off;
    const float* stream_mem = h16 ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(d->stream_mem) + mem_off) : d->stream_mem + mem_off;
    const bool one = nsub > 1;          // sub-run: one stream, M rows = R
    TV x = xq;
    for (int i = 0; i < d->n_layers; ++i) {
      const int lw = i * LBM_W_LAYER_COUNT;
      const int k = d->topk[i];
      TV lb = view("lb", k == 1 ? "lb_s" : nullptr, (int)lay.kmax, k, LBM_C);   // split twin only where a GEMM consumes it
      // collision then streaming read-out of the two memories (transformer.py:257-302)
      // layer 0 reads the (constant) query: its queries u = G x + gb may be kept by the caller across frames
      float* u0 = (i == 0 && d->query_u0) ? d->query_u0 + (long long)lo * 1024 : nullptr;
      if (attn_block(lw + LBM_W_LAYER_COLL, x, ta, coll_mem, (long long)M * LBM_C, M, d->ring_head, true, nullptr, nullptr,
                     (u0 && d->query_u0_valid) ? u0 : nullptr, 1024, (u0 && !d->query_u0_valid) ? u0 : nullptr))
        return -1;
      if (attn_block(lw + LBM_W_LAYER_STREAM, ta, tb, stream_mem, (long long)M * LBM_C, M, d->ring_head, true, &qn)) return -1;
      // correlation + top-k (transformer.py:306-308)
      float* cdst = d->corr[i] ? d->corr[i] + (long long)lo * lay.P0 : corr;
      const int gm = one ? R : d->N, gb = one ? 1 : d->B;
      bool fused_topk = false;
      int n_cand = 0;
      if (!split) {
        GemmArgs g;
        g.X = qn.f; g.ldx = LBM_C; g.sX = (long long)d->N * LBM_C;
        g.W = F + (long long)i * d->B * lay.P0 * LBM_C; g.ldw = LBM_C; g.sW = lay.P0 * LBM_C;
        g.Y = cdst; g.ldy = (int)lay.P0; g.sY = (long long)d->N * lay.P0;
        g.M = gm; g.N = (int)lay.P0; g.K = LBM_C; g.batch = gb;
        if (gemm(g, s)) return -1;
      } else {
        // F holds, per layer, the hi planes of all streams followed by their lo planes
        const long long plane = (long long)d->B * lay.P0 * LBM_C;
        // nobody asked for the correlation map itself: let the product keep only each 128-column tile's k best per row
        // (in the space the map would have taken) and select among those -- the R x P0 map is neither written nor re-read
        // (measured: the 9-deep per-lane lists cost the tile epilogue ~25 us at 1024 rows, more than the map's round trip
        // saves there; from ~4000 rows on the product is throughput-bound and the fusion wins 4 % of the C5 step)
        // Row-streaming form (gemm_rs.hip): the MAP streams through LDS, 256 points at a time sit in registers, and every lane
        // keeps the k best cells of its point across all steps -- no scores leave the accumulators at all.
        int rs_cand = 0;
        if (g_rs_topk && cplanes == 2 && d->corr[i] == nullptr && k <= 9) {
          rs_cand = gemm_rs_topk_cands((int)lay.P0, gm, gb, k);
          if (rs_cand < k || rs_cand > 4096 || rs_cand % 4 != 0 || 2ll * rs_cand > lay.P0) rs_cand = 0;
        }
        fused_topk = rs_cand > 0 || (cplanes == 2 && d->corr[i] == nullptr && gemm_ss_topk_ok(gm, (int)lay.P0, LBM_C, gb, k) &&
                                     (k <= g_fuse_kmax_small || Rfull >= 4096));
        GemmSSArgs g;
        g.planes = cplanes;
        if (rs_cand > 0) {
          n_cand = rs_cand;
          g.cand_v = cdst;
          g.cand_i = reinterpret_cast<int*>(cdst + (long long)R * n_cand);
          g.topk_k = k;
        } else if (fused_topk) {
          n_cand = (int)((lay.P0 + 127) / 128) * k;
          g.cand_v = cdst;
          g.cand_i = reinterpret_cast<int*>(cdst + (long long)R * n_cand);
          g.topk_k = k;
        }
        g.A = qn.s; g.a_plane = qn.plane; g.lda = LBM_C; g.sA = (long long)d->N * LBM_C;
        g.B = (const char*)F + (long long)i * plane * 4; g.b_plane = plane; g.ldb = LBM_C; g.sB = lay.P0 * LBM_C;
        g.Y = cdst; g.ldy = (int)lay.P0; g.sY = (long long)d->N * lay.P0;
        g.M = gm; g.N = (int)lay.P0; g.K = LBM_C; g.batch = gb;
        if (rs_cand > 0 ? gemm_rs_topk(g, g_rs_dbg, s) : gemm_ss(g, s)) return -1;
      }
      int* idst = d->topk_idx[i] ? d->topk_idx[i] + (long long)lo * k : idx;
      float* rdst = d->ref_points[i] ? d->ref_points[i] + (long long)lo * k * 2 : ref;
      if (fused_topk) {
        if (topk_candidates(cdst, reinterpret_cast<const int*>(cdst + (long long)R * n_cand), n_cand, (int)lay.P0, k, R, w, d->stride, idst,
                            rdst, nullptr, s))
          return -1;
      } else if (topk_rows(cdst, lay.P0, (int)lay.P0, k, R, w, d->stride, idst, rdst, nullptr, s)) {
        return -1;
      }
      // lattice deformable attention on the k regions (transformer.py:316-327)
      // the lattice offsets/weights and (k > 1) the update attention's queries are both products of the stream output:
      // one merged product [R, 48 P + 1024]
      TV olg = view("offlog", nullptr, 1, 1, 48 * 16 + 1024);
      const int ncol_ol = 48 * d->n_points;
      const bool merged = k > 1 && W(lw + LBM_W_LAYER_OLG_W) != nullptr;
      const int ld_olg = ncol_ol + 1024;
      if (merged && linear(tb, LBM_C, lw + LBM_W_LAYER_OLG_W, lw + LBM_W_LAYER_OLG_B, olg, ld_olg, R, ld_olg, LBM_C)) return -1;
      if (msda_block(lw + LBM_W_LAYER_LATTICE, i, d->n_points, tb, rdst, k, lb, merged ? olg.f : nullptr, ld_olg)) return -1;
      if (d->ref_rho[i]) {   // [1,256] read-out, kept on the exact fp32 path in both modes (weights stay fp32)
        GemmArgs g;
        g.X = lb.f; g.ldx = LBM_C; g.W = W(lw + LBM_W_LAYER_RHO_W); g.ldw = LBM_C; g.bias = W(lw + LBM_W_LAYER_RHO_B);
        g.Y = d->ref_rho[i] + (long long)lo * k; g.ldy = 1; g.M = R * k; g.N = 1; g.K = LBM_C;
        if (gemm(g, s)) return -1;
      }
      // lattice update (transformer.py:330-338)
      const TV& xn = (i & 1) ? x1 : x0;
      if (attn_block(lw + LBM_W_LAYER_UPDATE, tb, xn, lb.f, (long long)k * LBM_C, k, 0, false, nullptr, &lb,
                     merged ? olg.f + ncol_ol : nullptr, ld_olg))
        return -1;
      x = xn;
      if (i == d->n_layers - 1) ref = rdst;     // k = 1: the heads read the single cell centre where top-k left it
    }
    // ---- heads (transformer.py:103-149); `ref` now holds the last layer's single cell centre per point ----
    TV hidR = view("hidR", nullptr, 1, 1, LBM_C);
    float* pred = buf<float>("pred") + (long long)lo * 2;
    TV coll_tok = view("coll_tok", nullptr, 1, 1, LBM_C);
    float* offset = d->offset + (long long)lo * 2;
    // the three heads' deformable offsets/weights are products of the same token: one merged product
    TV ol3 = view("offlog", nullptr, 1, 1, 48 * 16 + 1024);
    const int c_off = 0, c_cm = 48 * d->n_points, c_vis = c_cm + 48 * d->n_points_cmem, ld3 = c_vis + 48 * d->n_points;
    const bool m3 = W(hb + LBM_W_HEAD_OL3_W) != nullptr && ld3 <= 48 * 16 + 1024;
    if (m3 && linear(x, LBM_C, hb + LBM_W_HEAD_OL3_W, hb + LBM_W_HEAD_OL3_B, ol3, ld3, R, ld3, LBM_C)) return -1;
    float* hid2 = nullptr;
    const int paired = heads_pair(hb, x, ref, m3 ? ol3.f : nullptr, c_off, c_vis, ld3, &hid2);
    if (paired < 0) return -1;
    const float* vis_hidden = nullptr;
    if (paired) {
      if (offset_final(hid2, W(hb + LBM_W_HEAD_OFFH_W3), W(hb + LBM_W_HEAD_OFFH_B3), ref, (float)d->stride, offset, pred, R, s)) return -1;
      if (msda_block(hb + LBM_W_HEAD_CMEM, d->n_layers + 1, d->n_points_cmem, x, pred, 1, coll_tok, m3 ? ol3.f + c_cm : nullptr, ld3)) return -1;
      vis_hidden = hid2 + (long long)R * LBM_C;
    } else {
      if (msda_block(hb + LBM_W_HEAD_OFFSET, d->n_layers, d->n_points, x, ref, 1, ta, m3 ? ol3.f + c_off : nullptr, ld3)) return -1;
      if (linear(ta, LBM_C, hb + LBM_W_HEAD_OFFH_W0, hb + LBM_W_HEAD_OFFH_B0, hidR, LBM_C, R, LBM_C, LBM_C, ACT_GELU)) return -1;
      if (offset_final(hidR.f, W(hb + LBM_W_HEAD_OFFH_W3), W(hb + LBM_W_HEAD_OFFH_B3), ref, (float)d->stride, offset, pred, R, s)) return -1;
      if (msda_block(hb + LBM_W_HEAD_CMEM, d->n_layers + 1, d->n_points_cmem, x, pred, 1, coll_tok, m3 ? ol3.f + c_cm : nullptr, ld3)) return -1;
      if (msda_block(hb + LBM_W_HEAD_VIS, d->n_layers + 2, d->n_points, x, ref, 1, ta, m3 ? ol3.f + c_vis : nullptr, ld3)) return -1;
      if (linear(ta, LBM_C, hb + LBM_W_HEAD_VISH_W0, hb + LBM_W_HEAD_VISH_B0, hidR, LBM_C, R, LBM_C, LBM_C, ACT_GELU)) return -1;
      vis_hidden = hidR.f;
    }
    StateFinalArgs f;
    f.vis_hidden = vis_hidden; f.w3 = W(hb + LBM_W_HEAD_VISH_W3); f.b3 = W(hb + LBM_W_HEAD_VISH_B3);
    f.stream_tok = x.f; f.coll_tok = coll_tok.f; f.query_init = query; f.queried = d->queried ? d->queried + lo : nullptr;
    f.stream_out = d->stream_out ? d->stream_out + (long long)lo * LBM_C : nullptr;
    f.coll_out = d->coll_out ? d->coll_out + (long long)lo * LBM_C : nullptr;
    if (d->update_state) {
      f.stream_mem = const_cast<float*>(stream_mem); f.coll_mem = const_cast<float*>(coll_mem);
      f.mem_f16 = h16 ? 1 : 0;
      f.vis_mask = d->vis_mask + (long long)lo * M; f.mem_mask = d->mem_mask + (long long)lo * M;
    }
    f.M = M; f.ring_head = d->ring_head; f.vis_threshold = d->vis_threshold;
    f.vis = d->vis + lo; f.rho = d->rho + lo; f.ref = ref; f.offset = offset; f.coords = d->coords ? d->coords + (long long)lo * 2 : nullptr;
    f.net_w = (float)d->net_w; f.net_h = (float)d->net_h; f.img_w = d->img_w; f.img_h = d->img_h;
    f.vis_flag = d->vis_flag ? d->vis_flag + lo : nullptr;
    f.rho_prob = d->rho_prob ? d->rho_prob + lo : nullptr;
    f.R = R;
    return state_final(f, s);
  }
};

}  // namespace lbm

namespace lbm { extern int g_corr_variant; extern int g_corr_skip_norm; extern int g_ln_serial; extern int g_ffn_claim; extern float* g_corr_trace; }
using namespace lbm;

extern "C" {

int lbm_hip_version(void) { return LBM_HIP_ABI_VERSION; }
const char* lbm_hip_last_error(void) { return g_err; }

int lbm_hip_linear(const lbm_hip_linear_desc* d, lbm_stream_t stream) {
  LBM_REQUIRE(d != nullptr, "linear: null descriptor");
  if (d->M == 0) return 0;
  LBM_REQUIRE(d->X && (d->W || (d->W_hi && d->W_lo)) && d->Y, "linear: null argument");
  GemmArgs a;
  a.X = d->X; a.W = d->W; a.bias = d->bias; a.Y = d->Y; a.res = d->res;
  a.M = d->M; a.N = d->N; a.K = d->K; a.ldx = d->ldx; a.ldw = d->ldw; a.ldy = d->ldy; a.ldres = d->ldres;
  a.res_div = d->res_div > 0 ? d->res_div : 1;
  a.res_mod = d->res_mod > 0 ? d->res_mod : 0x7fffffff;
  a.res_scale = d->res_scale; a.act = d->act; a.batch = d->batch > 0 ? d->batch : 1;
  a.sX = d->sX; a.sW = d->sW; a.sB = d->sB; a.sY = d->sY; a.sR = d->sR;
  a.ln_gamma = d->ln_gamma; a.ln_beta = d->ln_beta; a.keep = d->keep; a.fallback = d->fallback; a.Yn = d->Yn;
  if (d->W_hi && d->W_lo) {
    a.Wh = d->W_hi; a.Wl = d->W_lo;
    return gemm16(a, (hipStream_t)stream);
  }
  return gemm(a, (hipStream_t)stream);
}

int lbm_hip_linear_ss(const lbm_hip_linear_ss_desc* d, lbm_stream_t stream) {
  LBM_REQUIRE(d != nullptr, "linear_ss: null descriptor");
  GemmSSArgs a;
  a.A = d->A; a.B = d->B; a.a_plane = d->a_plane; a.b_plane = d->b_plane; a.bias = d->bias; a.Y = d->Y; a.Ys = d->Ys;
  a.ys_plane = d->ys_plane; a.res = d->res; a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldb = d->ldb; a.ldy = d->ldy;
  a.ldys = d->ldys; a.ldres = d->ldres; a.res_div = d->res_div > 0 ? d->res_div : 1;
  a.res_mod = d->res_mod > 0 ? d->res_mod : 0x7fffffff; a.res_scale = d->res_scale; a.act = d->act;
  a.batch = d->batch > 0 ? d->batch : 1; a.sA = d->sA; a.sB = d->sB; a.sBias = d->sBias; a.sY = d->sY; a.sYs = d->sYs; a.sR = d->sR;
  a.planes = d->planes == 1 ? 1 : 2;
  return gemm_ss(a, (hipStream_t)stream);
}

int lbm_hip_split_planes(const float* x, void* hi, void* lo, int64_t n, lbm_stream_t stream) {
  return split_planes(x, hi, lo, n, (hipStream_t)stream);
}

int lbm_hip_layernorm(const float* X, const float* gamma, const float* beta, float* Y, int32_t rows, const uint8_t* keep,
                      const float* fallback, float* Yn, lbm_stream_t stream) {
  return layernorm_rows(X, gamma, beta, Y, rows, keep, fallback, Yn, (hipStream_t)stream);
}

int lbm_hip_l2norm(const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t rows, lbm_stream_t stream) {
  return l2norm_rows(X, ldx, Y, ldy, rows, (hipStream_t)stream);
}

int lbm_hip_attn_readout(const float* U, const float* seq, int64_t point_stride, int64_t slot_stride, int32_t S,
                         int32_t ring_head, const float* temb, const uint8_t* mask_a, const uint8_t* mask_b, float* out,
                         uint8_t* valid, int32_t R, lbm_stream_t stream) {
  AttnArgs a;
  a.U = U; a.seq = seq; a.point_stride = point_stride; a.slot_stride = slot_stride; a.S = S; a.ring_head = ring_head;
  a.temb = temb; a.mask_a = mask_a; a.mask_b = mask_b; a.out = out; a.valid = valid; a.R = R;
  return attn_readout(a, (hipStream_t)stream);
}

int lbm_hip_attn_readout_h16(const float* U, const void* seq_f16, int64_t point_stride, int64_t slot_stride, int32_t S,
                             int32_t ring_head, const float* temb, const uint8_t* mask_a, const uint8_t* mask_b, float* out,
                             uint8_t* valid, int32_t R, lbm_stream_t stream) {
  AttnArgs a;
  a.U = U; a.seq = (const float*)seq_f16; a.seq_f16 = 1; a.point_stride = point_stride; a.slot_stride = slot_stride; a.S = S; a.ring_head = ring_head;
  a.temb = temb; a.mask_a = mask_a; a.mask_b = mask_b; a.out = out; a.valid = valid; a.R = R;
  return attn_readout(a, (hipStream_t)stream);
}

int lbm_hip_pyramid(float* tokens, int32_t B, int32_t h, int32_t w, lbm_stream_t stream) {
  return pyramid_pool(tokens, B, h, w, (hipStream_t)stream);
}

int lbm_hip_corr_maps(const float* tokens, float* inv_norm, float* F, const float* corr_mix_host, int32_t n_layers, int32_t B,
                      int32_t h, int32_t w, lbm_stream_t stream) {
  return corr_maps(tokens, inv_norm, F, corr_mix_host, n_layers, B, h, w, false, (hipStream_t)stream);
}

int lbm_hip_topk(const float* corr, int64_t ld, int32_t P, int32_t k, int32_t rows, int32_t grid_w, int32_t stride_px,
                 int32_t* idx_out, float* ref_out, float* val_out, lbm_stream_t stream) {
  return topk_rows(corr, ld, P, k, rows, grid_w, stride_px, idx_out, ref_out, val_out, (hipStream_t)stream);
}

int64_t lbm_hip_corr_topk_workspace_bytes(int32_t B, int32_t N, int32_t P0, int32_t k) {
  if (B < 1 || N < 1 || P0 < 1 || k < 1 || k > 9) return -1;
  const int nc = gemm_rs_topk_cands(P0, N, B, k);
  return nc < k ? -1 : (int64_t)B * N * nc * 8;
}

int lbm_hip_corr_topk(const void* q_hi, int64_t q_plane, const void* f_hi, int64_t f_plane, int32_t B, int32_t N, int32_t P0, int32_t k,
                      int32_t grid_w, int32_t stride_px, void* workspace, int64_t workspace_bytes, int32_t* idx_out, float* ref_out,
                      float* val_out, lbm_stream_t stream) {
  LBM_REQUIRE(q_hi && f_hi && workspace && idx_out, "corr_topk: null argument");
  LBM_REQUIRE(B >= 1 && N >= 1 && P0 >= k && k >= 1 && k <= 9 && grid_w >= 1, "corr_topk: bad shape B=%d N=%d P0=%d k=%d", B, N, P0, k);
  LBM_REQUIRE(q_plane % 8 == 0 && f_plane % 8 == 0 && ((uintptr_t)q_hi & 15) == 0 && ((uintptr_t)f_hi & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
              "corr_topk: operands must be 16-byte aligned");
  const int nc = gemm_rs_topk_cands(P0, N, B, k);
  LBM_REQUIRE(nc >= k && nc <= 4096 && nc % 4 == 0, "corr_topk: cannot partition %d cells (candidates %d)", P0, nc);
  const long long R = (long long)B * N;
  LBM_REQUIRE(workspace_bytes >= R * nc * 8, "corr_topk: workspace of %lld bytes, need %lld", (long long)workspace_bytes, R * nc * 8);
  GemmSSArgs g;
  g.A = q_hi; g.a_plane = q_plane; g.lda = LBM_C; g.sA = (long long)N * LBM_C;
  g.B = f_hi; g.b_plane = f_plane; g.ldb = LBM_C; g.sB = (long long)P0 * LBM_C;
  g.M = N; g.N = P0; g.K = LBM_C; g.batch = B;
  g.cand_v = static_cast<float*>(workspace);
  g.cand_i = reinterpret_cast<int*>(g.cand_v + R * nc);
  g.topk_k = k;
  if (gemm_rs_topk(g, g_rs_dbg, (hipStream_t)stream)) return -1;
  return topk_candidates(g.cand_v, g.cand_i, nc, P0, k, (int)R, grid_w, stride_px, idx_out, ref_out, val_out, (hipStream_t)stream);
}

int lbm_hip_msda_sample(const lbm_hip_msda_desc* d, lbm_stream_t stream) {
  LBM_REQUIRE(d != nullptr, "msda_sample: null descriptor");
  MsdaArgs a;
  a.value = d->value; a.v_batch_stride = d->v_batch_stride; a.ldv = d->ldv;
  for (int l = 0; l < 4; ++l) {
    a.lvl_h[l] = d->lvl_h[l]; a.lvl_w[l] = d->lvl_w[l]; a.lvl_start[l] = d->lvl_start[l];
  }
  a.B = d->B; a.N = d->N; a.k = d->k; a.P = d->P;
  a.offlog = d->offlog; a.ld_offlog = d->ld_offlog; a.logit_col = d->logit_col;
  a.ref = d->ref; a.net_w = d->net_w; a.net_h = d->net_h; a.loc = d->loc; a.attw = d->attw; a.out = d->out;
  a.value_f16 = d->value_f16 ? 1 : 0;
  return msda_sample(a, (hipStream_t)stream);
}

int lbm_hip_preprocess(const float* frames, int64_t frame_bstride, float* out, int32_t B, int32_t H, int32_t W, int32_t oh,
                       int32_t ow, int32_t out_nhwc, lbm_stream_t stream) {
  return preprocess(frames, frame_bstride, out, B, H, W, oh, ow, out_nhwc, (hipStream_t)stream);
}

int lbm_hip_preprocess_u8(const uint8_t* frames, int64_t frame_bstride, int32_t in_layout, float* out, int32_t B, int32_t H, int32_t W,
                          int32_t oh, int32_t ow, int32_t out_nhwc, lbm_stream_t stream) {
  return preprocess_u8(frames, frame_bstride, in_layout, out, B, H, W, oh, ow, out_nhwc, (hipStream_t)stream);
}

int lbm_hip_preprocess_split(const void* frames, int32_t is_u8, int64_t frame_bstride, int32_t in_layout, void* out_hi, int64_t out_plane,
                             int32_t B, int32_t H, int32_t W, int32_t oh, int32_t ow, lbm_stream_t stream) {
  return preprocess_split(frames, is_u8, frame_bstride, in_layout, out_hi, out_plane, B, H, W, oh, ow, (hipStream_t)stream);
}

int lbm_hip_stem_pool(const void* X, int64_t x_plane, const void* W_hi, const void* W_lo, int32_t ldw, const float* bias, float* Y, void* Ys,
                      int64_t ys_plane, int32_t B, int32_t H, int32_t W, lbm_stream_t stream) {
  return stem_pool(X, x_plane, W_hi, W_lo, ldw, bias, Y, Ys, ys_plane, B, H, W, (hipStream_t)stream);
}

int lbm_hip_upcat(const float* p1, const float* p2, const float* p3, float* out, int32_t B, int32_t h, int32_t w,
                  lbm_stream_t stream) {
  return upcat(p1, p2, p3, out, B, h, w, (hipStream_t)stream);
}

int lbm_hip_sample_queries(const float* tokens, int64_t tok_bstride, int32_t ld, const float* pts, float* out, int32_t B,
                           int32_t N, int32_t h, int32_t w, int32_t net_h, int32_t net_w, lbm_stream_t stream) {
  return sample_queries(tokens, tok_bstride, ld, pts, out, B, N, h, w, net_h, net_w, (hipStream_t)stream);
}

int lbm_hip_conv2d_nhwc(const float* X, const void* W, const float* bias, const float* res, float* Y, const void* W_lo,
                        int32_t B, int32_t H, int32_t Wd, int32_t Cin, int32_t Cout, int32_t kh, int32_t kw, int32_t stride,
                        int32_t pad, int32_t relu, int32_t ldw, lbm_stream_t stream) {
  ConvArgs c;
  c.X = X; c.bias = bias; c.res = res; c.Y = Y;
  if (W_lo) { c.Wh = W; c.Wl = W_lo; } else { c.W = (const float*)W; }
  c.B = B; c.H = H; c.Wd = Wd; c.Cin = Cin; c.Cout = Cout; c.kh = kh; c.kw = kw; c.stride = stride; c.pad = pad; c.relu = relu;
  c.ldw = ldw > 0 ? ldw : (kh * kw * Cin + 31) / 32 * 32;
  return W_lo ? conv2d_nhwc16(c, (hipStream_t)stream) : conv2d_nhwc(c, (hipStream_t)stream);
}

int lbm_hip_maxpool3x3s2_nhwc(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t C, lbm_stream_t stream) {
  return maxpool3x3s2_nhwc(x, y, B, H, W, C, (hipStream_t)stream);
}

int lbm_hip_debug_set(int32_t key, int32_t value) {
#ifndef LBM_TUNING
  // The timing-only ablation switches (results are wrong when set) are compiled into the tuning build only
  // (`make tuning`, -DLBM_TUNING): the product library refuses them instead of silently doing nothing.
  const bool ablation = (key == 14 || key == 16 || key == 20 || key == 23) ? value != 0
                        : key == 26                           ? (value & 255) != 0
                        : key == 18                           ? (value & 255) != 0
                        : key == 24                           ? (value & ~15) != 0
                        : key == 2                            ? (value & ~0x7f0) != 0
                                                              : false;
  if (ablation) {
    set_error("debug_set: key %d value %d selects timing-only ablation code, which this library was built without (make tuning)", key, value);
    return -1;
  }
#endif
  {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    g_knobs[key] = value;
  }
  if (key == 9) {
    lbm::g_corr_variant = value;
    return 0;
  }
  if (key == 10) {
    lbm::g_corr_skip_norm = value;
    return 0;
  }
  if (key == 11) {
    lbm::g_ln_serial = value;
    return 0;
  }
  if (key == 12) {
    lbm::g_ffn_claim = value;
    return 0;
  }
  if (key == 1) {
    gemm16_force_tile(value);
    return 0;
  }
  if (key == 2) {
    gemm16_debug(value);
    return 0;
  }
  if (key == 3) {
    attn_force_mode(value);
    return 0;
  }
  if (key == 4) {
    conv16_force(value);
    return 0;
  }
  if (key == 5) {
    topk_force_mode(value);
    return 0;
  }
  if (key == 6) {
    g_fuse_kmax_small = value;
    return 0;
  }
  if (key == 13) {
    g_rs_topk = value;
    return 0;
  }
  if (key == 15) {
    g_ffn_rows_mode = value;
    return 0;
  }
  if (key == 17) {
    g_head_pair = value;
    return 0;
  }
  if (key == 20) {
    lbm::g_stem_dbg = value;
    return 0;
  }
  if (key == 21) {
    lbm::g_msda_xcd = value;
    return 0;
  }
  if (key == 22) {
    lbm::g_msda_window = value;
    return 0;
  }
  if (key == 23) {
    lbm::g_msda_dbg = value;
    return 0;
  }
  if (key == 24) {
    lbm::g_tail_order = value;
    return 0;
  }
  if (key == 26) {
    lbm::g_conv_patch_dbg = value & 255;
    lbm::g_conv_patch_tile = value >> 8;
    return 0;
  }
  if (key == 25) {
    if (value != 3 && value != 4) {
      set_error("debug_set: key 25 (gemm_rs ring stages) takes 3 or 4");
      return -1;
    }
    lbm::g_rs_ring = value;
    return 0;
  }
  if (key == 18) {
    lbm::g_conv64_dbg = value;
    return 0;
  }
  if (key == 16) {
    lbm::g_ffn_rows_dbg = value;
    return 0;
  }
  if (key == 14) {
    g_rs_dbg = value;
    return 0;
  }
  if (key == 7) {
    g_ffn_fused_rows = value;
    return 0;
  }
  if (key == 8) {
    g_ffn_sites = value;
    return 0;
  }
  set_error("debug_set: unknown key %d", key);
  return -1;
}

int32_t lbm_hip_debug_active(int32_t* keys, int32_t* values, int32_t cap) {
  std::lock_guard<std::mutex> lk(g_knob_mu);
  int32_t n = 0;
  for (const auto& kv : g_knobs) {
    if (kv.second == knob_default(kv.first)) continue;
    if (n < cap && keys && values) {
      keys[n] = kv.first;
      values[n] = kv.second;
    }
    ++n;
  }
  return n;
}

int32_t lbm_hip_tuning_build(void) {
#ifdef LBM_TUNING
  return 1;
#else
  return 0;
#endif
}

int lbm_hip_debug_ptr(int32_t key, void* p) {
  if (key == 1) {
    lbm::g_corr_trace = (float*)p;
    return 0;
  }
  set_error("debug_ptr: unknown key %d", key);
  return -1;
}

int lbm_hip_profile_begin(void) {
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_prof.clear();
  g_prof_on = true;
  return 0;
}

int lbm_hip_profile_end(lbm_hip_profile_entry* out, int32_t cap) {
  g_prof_on = false;
  std::map<std::string, lbm_hip_profile_entry> agg;
  std::vector<std::string> order;
  for (auto& r : g_prof) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) {
      set_error("profile_end: event read failed");
      return -1;
    }
    auto it = agg.find(r.name);
    if (it == agg.end()) {
      lbm_hip_profile_entry e{};
      snprintf(e.name, sizeof(e.name), "%s", r.name);
      it = agg.emplace(r.name, e).first;
      order.push_back(r.name);
    }
    it->second.launches += 1;
    it->second.total_ms += ms;
    it->second.flops += r.flops;
    it->second.bytes += r.bytes;
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_prof.clear();
  int n = 0;
  for (auto& k : order) {
    if (out && n < cap) out[n] = agg[k];
    ++n;
  }
  return n;
}

int lbm_hip_conv2d_nhwc_ss(const void* X, int64_t x_plane, const void* W_hi, const void* W_lo, const float* bias, const float* res,
                           float* Y, void* Ys, int64_t ys_plane, const void* zeros, void* scratch, uint64_t scratch_bytes, int32_t B,
                           int32_t H, int32_t Wd, int32_t Cin, int32_t Cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad,
                           int32_t relu, int32_t ldw, lbm_stream_t stream) {
  ConvSSArgs c;
  c.X = X; c.x_plane = x_plane; c.Wh = W_hi; c.Wl = W_lo; c.bias = bias; c.res = res; c.Y = Y; c.Ys = Ys; c.ys_plane = ys_plane;
  c.zeros = zeros; c.scratch = scratch; c.scratch_bytes = scratch_bytes;
  c.B = B; c.H = H; c.Wd = Wd; c.Cin = Cin; c.Cout = Cout; c.kh = kh; c.kw = kw; c.stride = stride; c.pad = pad; c.relu = relu;
  c.ldw = ldw > 0 ? ldw : kh * kw * Cin;
  c.planes = W_lo == nullptr ? 1 : 2;
  return conv2d_nhwc_ss(c, (hipStream_t)stream);
}

int lbm_hip_maxpool3x3s2_nhwc_s(const float* x, float* y, void* ys, int64_t ys_plane, int32_t B, int32_t H, int32_t W, int32_t C,
                                lbm_stream_t stream) {
  return maxpool3x3s2_nhwc(x, y, B, H, W, C, (hipStream_t)stream, ys, ys_plane);
}

int lbm_hip_upcat_s(const float* p1, const float* p2, const float* p3, float* out, void* outs, int64_t outs_plane, int32_t B,
                    int32_t h, int32_t w, lbm_stream_t stream) {
  return upcat(p1, p2, p3, out, B, h, w, (hipStream_t)stream, outs, outs_plane);
}

int lbm_hip_pyramid_s(float* tokens, void* tokens_s, int64_t tokens_plane, int32_t B, int32_t h, int32_t w, lbm_stream_t stream) {
  return pyramid_pool(tokens, B, h, w, (hipStream_t)stream, tokens_s, tokens_plane);
}

int lbm_hip_encoder_tail(const float* q1, const float* q2, const float* q3, const float* pos, float* tokens, void* tokens_s,
                         int64_t tokens_plane, int32_t B, int32_t h, int32_t w, lbm_stream_t stream) {
  return encoder_tail(q1, q2, q3, pos, tokens, B, h, w, (hipStream_t)stream, tokens_s, tokens_plane);
}

int lbm_hip_encoder_tail_fused(const void* f1, int64_t f1_plane, const void* W1_hi, const void* W1_lo, int32_t ldw1, const float* b1,
                               const float* q2, const float* q3, const float* pos, float* tokens, void* tokens_s, int64_t tokens_plane,
                               float* inv_norm, int32_t B, int32_t h, int32_t w, lbm_stream_t stream) {
  return encoder_tail_fused(f1, f1_plane, W1_hi, W1_lo, ldw1, b1, q2, q3, pos, tokens, B, h, w, (hipStream_t)stream, tokens_s, tokens_plane,
                            inv_norm);
}

int lbm_hip_pack_fragment_major(const void* hi, const void* lo, int32_t N, int32_t K, void* out, lbm_stream_t stream) {
  return pack_fragment_major(hi, lo, N, K, out, (hipStream_t)stream);
}

int lbm_hip_ffn_slabs(const void* A, int64_t a_plane, int32_t lda, const void* F1_fm, const float* b1, const void* F2_fm, float* slabs,
                      int64_t slab_stride, int32_t M, int32_t planes, lbm_stream_t stream) {
  FfnArgs f;
  f.A = A; f.a_plane = a_plane; f.lda = lda; f.F1 = F1_fm; f.b1 = b1; f.F2 = F2_fm; f.slabs = slabs; f.slab_stride = slab_stride;
  f.M = M; f.chunk = 128; f.planes = planes == 1 ? 1 : 2;
  if (f.planes == 2 && g_ffn_rows_mode != 1 && (g_ffn_rows_mode == 2 || M > g_ffn_fused_rows)) return ffn_rows(f, (hipStream_t)stream);
  return ffn_fused(f, (hipStream_t)stream);
}

int lbm_hip_update_points(const uint8_t* remove_mask, int32_t N_old, const uint8_t* update_mask, int32_t N_new, int32_t M,
                          const float* old_query, const float* old_stream, const float* old_coll, const uint8_t* old_vis,
                          const uint8_t* old_mem, const float* new_query, float* query, float* stream_mem, float* coll_mem,
                          uint8_t* vis_mask, uint8_t* mem_mask, int32_t* scratch, lbm_stream_t stream) {
  UpdatePointsArgs a;
  a.remove_mask = remove_mask; a.update_mask = update_mask; a.N_old = N_old; a.N_new = N_new; a.M = M;
  a.old_query = old_query; a.old_stream = old_stream; a.old_coll = old_coll; a.old_vis = old_vis; a.old_mem = old_mem;
  a.new_query = new_query; a.query = query; a.stream = stream_mem; a.coll = coll_mem; a.vis = vis_mask; a.mem = mem_mask;
  a.scratch = scratch;
  return update_points(a, (hipStream_t)stream);
}

int lbm_hip_update_points_static(const uint8_t* remove_mask, int32_t N_old, const uint8_t* update_mask, int32_t N_new, int32_t M,
                                 const float* old_query, const float* old_stream, const float* old_coll, const uint8_t* old_vis,
                                 const uint8_t* old_mem, const float* new_query, float* query, float* stream_mem, float* coll_mem,
                                 uint8_t* vis_mask, uint8_t* mem_mask, lbm_stream_t stream) {
  UpdatePointsArgs a;
  a.remove_mask = remove_mask; a.update_mask = update_mask; a.N_old = N_old; a.N_new = N_new; a.M = M;
  a.old_query = old_query; a.old_stream = old_stream; a.old_coll = old_coll; a.old_vis = old_vis; a.old_mem = old_mem;
  a.new_query = new_query; a.query = query; a.stream = stream_mem; a.coll = coll_mem; a.vis = vis_mask; a.mem = mem_mask;
  return update_points_static(a, (hipStream_t)stream);
}

uint64_t lbm_hip_step_workspace_bytes(const lbm_hip_step_desc* d) {
  StepLayout lay;
  if (build_layout(d, lay)) return 0;
  return lay.total;
}

int64_t lbm_hip_step_workspace_offset(const lbm_hip_step_desc* d, const char* name) {
  StepLayout lay;
  if (build_layout(d, lay) || name == nullptr) return -1;
  return lay.find(name);
}

int lbm_hip_track_step(const lbm_hip_step_desc* d, lbm_stream_t stream) {
  Step st;
  if (build_layout(d, st.lay)) return -1;
  LBM_REQUIRE(d->weights && d->n_weights == d->n_layers * LBM_W_LAYER_COUNT + LBM_W_HEAD_COUNT,
              "step: weight table has %d entries, expected %d", d->n_weights, d->n_layers * LBM_W_LAYER_COUNT + LBM_W_HEAD_COUNT);
  LBM_REQUIRE(d->tokens && d->workspace, "step: null tokens/workspace");
  LBM_REQUIRE(d->workspace_bytes >= st.lay.total, "step: workspace has %llu bytes, needs %llu",
              (unsigned long long)d->workspace_bytes, (unsigned long long)st.lay.total);
  LBM_REQUIRE(((uintptr_t)d->workspace & 255) == 0, "step: workspace must be 256-byte aligned");
  LBM_REQUIRE(d->n_points >= 1 && d->n_points <= 16 && d->n_points_cmem >= 1 && d->n_points_cmem <= 16, "step: bad n_points");
  LBM_REQUIRE(d->ring_head >= 0 && d->ring_head < d->M, "step: ring_head %d outside [0,%d)", d->ring_head, d->M);
  if (st.lay.R > 0) {
    LBM_REQUIRE(d->query && d->stream_mem && d->coll_mem && d->vis_mask && d->mem_mask, "step: null state pointer");
    LBM_REQUIRE(d->vis && d->rho && d->offset, "step: vis/rho/offset outputs are required");
  }
  st.split = d->precision >= 1;
  st.planes = d->precision >= 2 ? 1 : 2;
  st.cplanes = d->precision == 2 ? 1 : 2;
  st.h16 = d->precision == 3;
  if (d->precision == 1) LBM_REQUIRE(d->weights_lo != nullptr, "step: precision 1 needs the lo planes of the weight matrices (weights_lo)");
  st.d = d;
  st.s = (hipStream_t)stream;
  st.ws = (char*)d->workspace;
  st.h = d->net_h / d->stride;
  st.w = d->net_w / d->stride;
  st.Rfull = st.lay.R;
  LBM_REQUIRE(d->phase >= 0 && d->phase <= 2, "step: phase must be 0, 1 or 2");
  if (d->phase == 2) {
    LBM_REQUIRE(d->row_lo >= 0 && d->row_count >= 0 && (long long)d->row_lo + d->row_count <= st.lay.R, "step: row range outside [0, %lld)", st.lay.R);
    LBM_REQUIRE(d->sub_index >= 0 && d->sub_index < 4, "step: sub_index must be in [0, 4)");
    const bool whole = d->row_lo == 0 && d->row_count == st.lay.R;
    LBM_REQUIRE(whole || d->B == 1, "step: a partial row range needs B == 1");
    st.lo = d->row_lo; st.R = d->row_count; st.sub = d->sub_index; st.nsub = whole ? 1 : 2;
    return st.run_points();
  }
  if (st.run_frame()) return -1;
  if (d->phase == 1) return 0;

  // Point-parallel sub-streams: with one stream of ~1000 points every kernel of the per-point chain is a small,
  // latency-bound launch; points are independent, so the rows are cut into up to 4 ranges whose chains run
  // concurrently on the caller's auxiliary HIP streams (fork after the frame phase, join at the end).
  int nsub = 1;
  if (d->n_aux_streams > 0 && d->B == 1) {
    LBM_REQUIRE(d->n_aux_streams <= 3, "step: at most 3 auxiliary streams");
    nsub = 1 + d->n_aux_streams;
    while (nsub > 1 && st.lay.R < 256ll * nsub) --nsub;
  }
  if (nsub == 1) {
    st.lo = 0; st.R = (int)st.lay.R; st.sub = 0; st.nsub = 1;
    return st.run_points();
  }
  // fork / join events: one set per (calling stream), created on first use under a lock, so that two host threads driving two
  // trackers on two streams never share (or race to create) an event
  struct AuxEvents { hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr}; };
  static std::mutex ev_mu;
  static std::map<hipStream_t, AuxEvents> ev_by_stream;
  hipEvent_t ev_fork, ev_join[3];
  {
    std::lock_guard<std::mutex> lk(ev_mu);
    AuxEvents& e = ev_by_stream[(hipStream_t)stream];
    if (e.fork == nullptr) {
      LBM_REQUIRE(hipEventCreateWithFlags(&e.fork, hipEventDisableTiming) == hipSuccess, "step: event creation failed");
      for (int i = 0; i < 3; ++i)
        LBM_REQUIRE(hipEventCreateWithFlags(&e.join[i], hipEventDisableTiming) == hipSuccess, "step: event creation failed");
    }
    ev_fork = e.fork;
    for (int i = 0; i < 3; ++i) ev_join[i] = e.join[i];
  }
  hipStream_t main_s = (hipStream_t)stream;
  LBM_REQUIRE(hipEventRecord(ev_fork, main_s) == hipSuccess, "step: fork record failed");
  const long long chunk = ((st.lay.R + nsub - 1) / nsub + 63) / 64 * 64;
  int rc = 0;
  for (int i = 0; i < nsub; ++i) {
    Step sub = st;
    sub.sub = i; sub.nsub = nsub;
    sub.lo = (int)std::min<long long>(st.lay.R, chunk * i);
    sub.R = (int)(std::min<long long>(st.lay.R, chunk * (i + 1)) - sub.lo);
    if (i > 0) {
      sub.s = (hipStream_t)d->aux_streams[i - 1];
      LBM_REQUIRE(sub.s != nullptr && sub.s != main_s, "step: auxiliary stream %d is null or equals the main stream", i - 1);
      LBM_REQUIRE(hipStreamWaitEvent(sub.s, ev_fork, 0) == hipSuccess, "step: fork wait failed");
    }
    if (rc == 0) rc = sub.run_points();
    if (i > 0) {   // always join, also after an error, so the streams stay consistent (and a capture stays closed)
      LBM_REQUIRE(hipEventRecord(ev_join[i - 1], sub.s) == hipSuccess, "step: join record failed");
      LBM_REQUIRE(hipStreamWaitEvent(main_s, ev_join[i - 1], 0) == hipSuccess, "step: join wait failed");
    }
  }
  return rc;
}

}  // extern "C"
