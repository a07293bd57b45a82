// xhist_plan.hip.h — host side: edge tables of a plan (build on device, verify on host), plan create / destroy / parameters
// Part of the single translation unit xhist_capi.hip (included there, in order).
#pragma once

// Build the device table blob of one compare domain:
//   [per-dimension edge arrays, 8-byte aligned] [per-dimension bucket tables (uint32 x K)]
// dom: 0 float64, 1 int64, 2 float32 thresholds.  `words[d]` holds dimension d's edge array
// already converted to the domain's element type; `edges` are the caller's original arrays.
// dim_dom (optional): per-dimension domain (0 / 1) overriding `dom` — plans with mixed domains
static int build_domain(xhist_plan* p, int dom_all, bool lut16, int n_inputs, const int64_t* n_edges,
                        const std::vector<std::vector<uint64_t>>& words, const void* const* edges, TableSet* ts,
                        const int* dim_dom = nullptr) {
  DimTable* dims = ts->dim;
  uint64_t** d_blob_out = &ts->blob;
  int32_t* table_words_out = &ts->words;
  int* max_cnt_out = &ts->max_cnt;
  int32_t edge_off = 0;
  int64_t max_e = 0;
  for (int d = 0; d < n_inputs; ++d) {
    DimTable& t = dims[d];
    memset(&t, 0, sizeof t);
    const int E = (int)n_edges[d];
    max_e = std::max<int64_t>(max_e, E);
    t.n_edges = E;
    t.nb = E - 1;
    t.edge_off = edge_off;
    edge_off += (int32_t)words[d].size();
    const int dom = dim_dom ? dim_dom[d] : dom_all;
    t.is_i64 = dim_dom && dom == 1;
    double range;
    if (dom == 0) {
      const double* e = static_cast<const double*>(edges[d]);
      t.e0_f = e[0];
      t.eL_f = e[E - 1];
      range = t.eL_f - t.e0_f;
    } else if (dom == 1) {
      const int64_t* e = static_cast<const int64_t*>(edges[d]);
      t.e0_i = e[0];
      t.eL_i = e[E - 1];
      range = (double)((uint64_t)t.eL_i - (uint64_t)t.e0_i);
    } else {
      const double* e = static_cast<const double*>(edges[d]);
      const float* thr = reinterpret_cast<const float*>(words[d].data());
      float last = (float)e[E - 1];  // largest float32 <= e_last
      if ((double)last > e[E - 1]) last = std::nextafterf(last, -INFINITY);
      t.e0_f = (double)thr[0];
      t.eL_f = (double)last;
      range = (double)((float)t.eL_f - (float)t.e0_f);
    }
    int K = std::min(4096, std::max(8, next_pow2((int)std::min<int64_t>(4 * (int64_t)E, 1 << 20))));
    if (lut16) K *= 2;  // 2-byte entries: twice the buckets for the same LDS bytes
    // more than 65535 edges: `start` no longer fits the 16-bit table fields — no bucket table at
    // all (lut_k = 0): digitize is a plain binary search over the edge array (generic family), or
    // table-free when the edges are arithmetic
    const bool no_lut = E > 65535;
    double scale = (double)K / range;
    if (dom == 2) scale = (double)(float)scale;
    if (!(range > 0.0) || !std::isfinite(range) || !std::isfinite(scale) || !(scale > 0.0)) {
      K = 1;  // degenerate span: one bucket holding every edge, pure binary search
      scale = 0.0;
    }
    if (no_lut) { K = 0; scale = 0.0; }
    t.lut_k = K;
    t.scale = scale;
    if (dom == 2) t.bias = (double)(-(float)t.e0_f * (float)scale);
    else if (dom == 0) t.bias = -t.e0_f * scale;
    if (!std::isfinite(t.bias)) {  // e.g. e_0 = -inf with scale 0: keep the map defined (bucket 0)
      t.bias = 0.0;
      if (K > 1) { K = 1; t.lut_k = 1; t.scale = 0.0; }
    }
  }
  int64_t stride = 1;
  for (int d = n_inputs - 1; d >= 0; --d) {
    dims[d].out_stride = stride;
    stride *= dims[d].nb;
  }
  // bucket tables follow the edges; lut_off counts table ENTRIES (4-byte, or 2-byte for lut16)
  int32_t off = (lut16 ? 4 : 2) * edge_off;
  for (int d = 0; d < n_inputs; ++d) {
    dims[d].lut_off = off;
    off += dims[d].lut_k;
  }
  const int32_t per_word = lut16 ? 4 : 2;
  const int32_t table_words = (off + per_word - 1) / per_word;
  std::vector<uint64_t> blob((size_t)table_words, 0);
  for (int d = 0; d < n_inputs; ++d) memcpy(blob.data() + dims[d].edge_off, words[d].data(), words[d].size() * 8);

  uint64_t* d_blob = nullptr;
  int32_t* d_scratch = nullptr;
  auto cleanup = [&](int rc) {
    if (d_scratch) (void)hipFree(d_scratch);
    if (rc != XHIST_OK && d_blob) (void)hipFree(d_blob);
    return rc;
  };
#define HIPP(expr)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return cleanup(fail(XHIST_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); \
  } while (0)
  HIPP(hipMalloc(&d_blob, blob.size() * 8));
  HIPP(hipMalloc(&d_scratch, (size_t)max_e * 4));
  HIPP(hipMemcpy(d_blob, blob.data(), blob.size() * 8, hipMemcpyHostToDevice));
  for (int d = 0; d < n_inputs; ++d) {
    if (dims[d].lut_k == 0) continue;
    const int dom = dim_dom ? dim_dom[d] : dom_all;
    HIPP((hipError_t)xhist_hot_build_tables(dom, lut16, dims[d], d_blob, d_scratch));  // (the builders live in the small hot code object: xhist_hot.hip)
    HIPP(hipDeviceSynchronize());
  }
  HIPP(hipMemcpy(blob.data(), d_blob, blob.size() * 8, hipMemcpyDeviceToHost));
#undef HIPP
  const uint32_t* lut4 = reinterpret_cast<const uint32_t*>(blob.data());
  const uint16_t* lut2 = reinterpret_cast<const uint16_t*>(blob.data());
  for (int d = 0; d < n_inputs; ++d) {
    DimTable& t = dims[d];
    uint32_t maxcnt = 0;
    uint64_t total = 0;
    if (t.lut_k == 0) { maxcnt = (uint32_t)t.n_edges; total = (uint64_t)t.n_edges; }
    for (int b = 0; b < t.lut_k; ++b) {
      uint32_t cnt;
      if (lut16) {
        const uint32_t next = b + 1 < t.lut_k ? lut2[t.lut_off + b + 1] : (uint32_t)t.n_edges;
        cnt = next - lut2[t.lut_off + b];
      } else {
        cnt = lut4[t.lut_off + b] >> 16;
      }
      maxcnt = std::max(maxcnt, cnt);
      total += cnt;
    }
    if (total != (uint64_t)t.n_edges) return cleanup(fail(XHIST_ERR_HIP, "bucket table of dim %d is inconsistent", d));
    *max_cnt_out = std::max<int>(*max_cnt_out, (int)maxcnt);
    int steps = 0;
    while ((1u << steps) <= maxcnt) ++steps;
    t.steps = steps;
  }
  (void)p;
  *d_blob_out = d_blob;
  *table_words_out = table_words;
  return cleanup(XHIST_OK);
}

// Packed-entry table set (count_le_pack; kernels header): per dimension K 16-byte entries behind the float64 edges.
// K is searched per dimension, downwards from what the LDS budget allows, for a bucket grid on which no bucket holds more
// than three edges (C3: 257 random edges, K ~ 850: most grids qualify, a few put four edges of a tight cluster into one
// bucket); the search runs on the host with the float32 arithmetic of bucket_of<2>, the table is BUILT on the device and
// read back, and only what the device built decides whether the set is offered.
// f32dom: the set for float32 SAMPLES (count_le_pack_f32) — thresholds are the smallest float32 >= e_j (> e_last for the
// last edge: the right-edge rule lives in the table), computed here; the blob holds entries only.
static int build_pack_domain(xhist_plan* p, bool f32dom, int n_inputs, const int64_t* n_edges, const std::vector<std::vector<uint64_t>>& words,
                             const void* const* edges, size_t entry_budget_bytes) {
  TableSet* ts = f32dom ? &p->ts_pk32 : &p->ts_pk;
  int& np_out = f32dom ? p->pk32_np : p->pk_np;
  np_out = 0;
  ts->max_cnt = 0;            // (a retry with a smaller budget runs on the same TableSet: nothing of the last attempt may survive)
  bool any_key_map = false;   // some dimension uses the float-bits map (a local, not a sentinel left in ts->max_cnt)
  std::vector<std::vector<float>> thr_all((size_t)n_inputs);
  int32_t edge_off = 0;
  int64_t max_e = 0;
  const int k_budget = (int)std::min<size_t>(entry_budget_bytes / (16 * (size_t)n_inputs), 2048);
  for (int d = 0; d < n_inputs; ++d) {
    DimTable& t = ts->dim[d];
    memset(&t, 0, sizeof t);
    const int E = (int)n_edges[d];
    const double* e = static_cast<const double*>(edges[d]);
    max_e = std::max<int64_t>(max_e, E);
    t.n_edges = E;
    t.nb = E - 1;
    t.e0_f = e[0];
    t.eL_f = e[E - 1];
    t.edge_off = edge_off;
    if (!f32dom) edge_off += (int32_t)words[d].size();
    if (E < 2 || E > 65535 || E > 3 * k_budget) return XHIST_OK;  // (more than three edges per bucket whatever the grid)
    // buckets: eight per edge are plenty (a finer grid separates nothing more that matters) — every workgroup stages the
    // table, and a 32 KB table behind 50 edges cost the many-small-workgroups shapes 30 % (456 rows x 10^6 float32, 50 random
    // edges: 0.377 ms against 0.29; profiles/r04_j_*)
    const int k_cap = std::min(k_budget, std::max(64, 8 * E));
    std::vector<float>& thr = thr_all[(size_t)d];
    thr.resize((size_t)E);
    for (int j = 0; j < E; ++j) {
      if (!std::isfinite(e[j]) || std::fabs(e[j]) > 3.0e38) return XHIST_OK;
      float f = (float)e[j];
      if (f32dom) {
        if (j < E - 1 ? (double)f < e[j] : (double)f <= e[j]) f = std::nextafterf(f, INFINITY);
      }
      thr[(size_t)j] = f;
    }
    const float range = thr[(size_t)E - 1] - thr[0];
    if (!(range > 0.0f) || !std::isfinite(range)) return XHIST_OK;
    int best_k = 0, best_cnt = 4;
    for (int K = k_cap; K >= std::max(8, k_cap - 96) && best_cnt > 2; --K) {
      const float scale = (float)((double)K / (double)range), bias = -thr[0] * scale;
      if (!std::isfinite(scale) || !(scale > 0.0f) || !std::isfinite(bias)) continue;
      int run = 0, prev = -1, mx = 0;
      for (int j = 0; j < E; ++j) {
        float tt = std::fmaf(thr[(size_t)j], scale, bias);
        tt = std::fmax(std::fmin(tt, (float)(K - 1)), 0.0f);
        const int b = (int)tt;
        run = b == prev ? run + 1 : 1;
        prev = b;
        mx = std::max(mx, run);
      }
      if (mx < best_cnt) { best_cnt = mx; best_k = K; }
    }
    if (best_k && best_cnt <= 3) {
      const float scale = (float)((double)best_k / (double)range);
      t.lut_k = best_k;
      t.scale = (double)scale;
      t.bias = (double)(-thr[0] * scale);
      t.steps = 1;
      continue;
    }
    // a linear grid cannot separate these edges (geometric / logarithmic spacing: most of them sit in its first buckets):
    // buckets on the float32 bit pattern instead — uniform in log x; integer arithmetic, the same on host and device
    // edges on both sides of zero: lift magnitudes below the smallest non-zero |edge| and cut the empty binades around zero out
    // of the key space (DimTable::key_floor) — a symmetric-log axis then has as many buckets per decade as a one-sided one
    float kfloor = 0.0f;
    uint32_t kpos0 = 0u, kgap = 0u;
    if (thr[0] < 0.0f && thr[(size_t)E - 1] > 0.0f) {
      float m = INFINITY;
      for (int j = 0; j < E; ++j)
        if (thr[(size_t)j] != 0.0f) m = std::min(m, std::fabs(thr[(size_t)j]));
      if (std::isfinite(m) && m > 0.0f) {
        kfloor = m;
        kpos0 = float_order_key(m);
        kgap = kpos0 - float_order_key(-m) - 1u;
      }
    }
    auto key_of = [&](float v) {  // (as bucket_of_key sees it, before the shift)
      if (kfloor > 0.0f) {
        v += 0.0f;
        v = std::copysign(std::fmax(std::fabs(v), kfloor), v);
      }
      uint32_t k = float_order_key(v);
      if (kfloor > 0.0f && k >= kpos0) k -= kgap;
      return k;
    };
    const uint32_t k0 = key_of(thr[0]), k1 = key_of(thr[(size_t)E - 1]);
    // buckets per edge the search stops at (development override: XHIST_AMD_KEY_K_PER_EDGE)
    static const double key_k_per_edge = [] { const char* e = getenv("XHIST_AMD_KEY_K_PER_EDGE"); return e && *e ? atof(e) : 0.0; }();
    int best_shift = -1;
    best_k = 0;
    for (int shift = 0; shift < 32; ++shift) {
      const uint64_t K64 = ((uint64_t)(k1 - k0) >> shift) + 1;
      if (K64 > (uint64_t)k_cap) continue;
      const int K = (int)std::max<uint64_t>(K64, 2);
      int run = 0, prev = -1, mx = 0;
      for (int j = 0; j < E; ++j) {
        const int b = bucket_of_key(thr[(size_t)j], k0, shift, K, kfloor, kpos0, kgap);
        run = b == prev ? run + 1 : 1;
        prev = b;
        mx = std::max(mx, run);
      }
      // the general kernels always compare three thresholds, so the COARSEST grid that keeps three edges per bucket is the
      // best one: the smallest table to stage and to hold in LDS (401 geometric edges: 192 buckets = 3 KB instead of 1536 =
      // 24 KB).  Coarser grids only merge buckets, so the search ends at the first one that holds four.
      if (mx > 3) break;
      best_shift = shift;
      best_k = K;
      if ((double)K <= key_k_per_edge * (double)E) break;  // coarse enough (see key_k_per_edge)
    }
    if (best_shift < 0) return XHIST_OK;
    t.lut_k = best_k;
    t.map_kind = 1;
    t.key_lo = k0;
    t.key_shift = best_shift;
    t.key_floor = kfloor;
    t.key_pos0 = kpos0;
    t.key_gap = kgap;
    t.steps = 1;
    any_key_map = true;
  }
  int64_t stride = 1;
  for (int d = n_inputs - 1; d >= 0; --d) {
    ts->dim[d].out_stride = stride;
    stride *= ts->dim[d].nb;
  }
  int32_t off = (edge_off + 1) / 2;  // entries: 16-byte units, behind the edges
  for (int d = 0; d < n_inputs; ++d) {
    ts->dim[d].lut_off = off;
    off += ts->dim[d].lut_k;
  }
  const int32_t table_words = off * 2;
  std::vector<uint64_t> blob((size_t)table_words, 0);
  if (!f32dom)
    for (int d = 0; d < n_inputs; ++d) memcpy(blob.data() + ts->dim[d].edge_off, words[d].data(), words[d].size() * 8);
  uint64_t* d_blob = nullptr;
  int32_t* d_scratch = nullptr;
  auto cleanup = [&](int rc, bool keep) {
    if (d_scratch) (void)hipFree(d_scratch);
    if (!keep && d_blob) (void)hipFree(d_blob);
    return rc;
  };
#define HIPP(expr)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return cleanup(fail(XHIST_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)), false); \
  } while (0)
  HIPP(hipMalloc(&d_blob, blob.size() * 8));
  HIPP(hipMalloc(&d_scratch, (size_t)max_e * 4 * 2));  // bucket ids, and (float32 domain) the thresholds behind them
  HIPP(hipMemcpy(d_blob, blob.data(), blob.size() * 8, hipMemcpyHostToDevice));
  for (int d = 0; d < n_inputs; ++d) {
    float* d_thr = nullptr;
    if (f32dom) {
      d_thr = reinterpret_cast<float*>(d_scratch + max_e);
      HIPP(hipMemcpy(d_thr, thr_all[(size_t)d].data(), thr_all[(size_t)d].size() * 4, hipMemcpyHostToDevice));
    }
    HIPP((hipError_t)xhist_hot_build_pack_tables(ts->dim[d], d_blob, d_scratch, (const float*)d_thr));
    HIPP(hipDeviceSynchronize());
  }
  HIPP(hipMemcpy(blob.data(), d_blob, blob.size() * 8, hipMemcpyDeviceToHost));
#undef HIPP
  int np = 1;
  const uint32_t* w32 = reinterpret_cast<const uint32_t*>(blob.data());
  for (int d = 0; d < n_inputs; ++d) {
    const DimTable& t = ts->dim[d];
    uint32_t prev = 0;
    for (int b = 0; b <= t.lut_k; ++b) {
      const uint32_t start = b < t.lut_k ? w32[((size_t)t.lut_off + (size_t)b) * 4 + 3] + 1u : (uint32_t)t.n_edges;  // (entries hold start - 1)
      if (start < prev || start > (uint32_t)t.n_edges || (b == 0 && start != 0)) return cleanup(XHIST_OK, false);  // (not offered)
      if (b) np = std::max<int>(np, (int)(start - prev));
      prev = start;
    }
  }
  if (np > 3) return cleanup(XHIST_OK, false);
  ts->blob = d_blob;
  ts->words = table_words;
  ts->max_cnt = np;
  np_out = any_key_map ? 4 : (np <= 2 ? 2 : 3);  // 4: the general kernels (map per dimension, three edges per bucket)
  return cleanup(XHIST_OK, true);
}

extern "C" int xhist_plan_create(int device, int n_inputs, const void* const* edges, const int64_t* n_edges,
                                 int cmp_domain, xhist_plan** out_plan) {
  Range range_("xhist_plan_create[edge tables]");
  if (!out_plan) return fail(XHIST_ERR_INVALID, "plan out-pointer is NULL");
  *out_plan = nullptr;
  if (n_inputs < 1 || n_inputs > XHIST_MAX_DIMS)
    return fail(XHIST_ERR_INVALID, "n_inputs must be in [1, %d], got %d", XHIST_MAX_DIMS, n_inputs);
  if (!edges || !n_edges) return fail(XHIST_ERR_INVALID, "edges / n_edges is NULL");
  // per-input domains: bit d of the mask = input d compares in int64; uniform masks are the plain domains
  int dim_dom[XHIST_MAX_DIMS] = {0};
  bool mixed = false;
  const bool uns = (cmp_domain & XHIST_CMP_UNSIGNED) != 0;  // int64-domain inputs hold unsigned values
  cmp_domain &= ~XHIST_CMP_UNSIGNED;
  if ((cmp_domain & ~0xff) == XHIST_CMP_PER_DIM) {
    const int mask = cmp_domain & 0xff;
    if (mask >> n_inputs) return fail(XHIST_ERR_INVALID, "per-input compare mask 0x%x names inputs beyond the %d given", mask, n_inputs);
    if (mask == 0) cmp_domain = XHIST_CMP_F64;
    else if (mask == (1 << n_inputs) - 1) cmp_domain = XHIST_CMP_I64;
    else mixed = true;
    for (int d = 0; d < n_inputs; ++d) dim_dom[d] = (mask >> d) & 1;
  } else if (cmp_domain != XHIST_CMP_F64 && cmp_domain != XHIST_CMP_I64) {
    return fail(XHIST_ERR_INVALID, "unknown compare domain %d", cmp_domain);
  }
  if (!mixed)
    for (int d = 0; d < n_inputs; ++d) dim_dom[d] = cmp_domain == XHIST_CMP_I64 ? 1 : 0;
  int64_t max_e = 0;
  for (int d = 0; d < n_inputs; ++d) {
    if (!edges[d]) return fail(XHIST_ERR_INVALID, "edges[%d] is NULL", d);
    if (n_edges[d] < 1) return fail(XHIST_ERR_INVALID, "edges[%d] needs at least one edge", d);
    if (n_edges[d] > ((int64_t)1 << 30))
      return fail(XHIST_ERR_UNSUPPORTED, "edges[%d] has %lld edges; this build supports at most 2^30 per dimension", d,
                  (long long)n_edges[d]);
    max_e = std::max(max_e, n_edges[d]);
    if (dim_dom[d] == 0) {
      const double* e = static_cast<const double*>(edges[d]);
      for (int64_t j = 0; j < n_edges[d]; ++j) {
        if (e[j] != e[j]) return fail(XHIST_ERR_EDGES, "edges[%d] contains NaN", d);
        if (j && e[j] < e[j - 1]) return fail(XHIST_ERR_EDGES, "bins must increase monotonically (edges[%d])", d);
      }
    } else if (uns) {
      const uint64_t* e = static_cast<const uint64_t*>(edges[d]);
      for (int64_t j = 1; j < n_edges[d]; ++j)
        if (e[j] < e[j - 1]) return fail(XHIST_ERR_EDGES, "bins must increase monotonically (edges[%d])", d);
    } else {
      const int64_t* e = static_cast<const int64_t*>(edges[d]);
      for (int64_t j = 1; j < n_edges[d]; ++j)
        if (e[j] < e[j - 1]) return fail(XHIST_ERR_EDGES, "bins must increase monotonically (edges[%d])", d);
    }
  }
  if (device < 0 || device >= n_devices())
    return fail(XHIST_ERR_NO_DEVICE, "HIP device %d not available (%d visible); this library has no CPU path", device,
                n_devices());
  DeviceGuard g;
  if (int rc = g.set(device)) return rc;

  xhist_plan* p = new (std::nothrow) xhist_plan();
  if (!p) return fail(XHIST_ERR_NOMEM, "out of host memory");
  p->device = device;
  p->n_dims = n_inputs;
  p->cmp = cmp_domain;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, physical_device(device)) == hipSuccess) {
    p->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    p->lds_max = prop.sharedMemPerBlock;
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, physical_device(device)) == hipSuccess && optin > 0)
      p->lds_max = std::max(p->lds_max, (size_t)optin);
  }

  // ---- table sets: one per compare domain this plan can be asked for -----------------------
  //   native (float64 or int64): every kernel family;  float32 thresholds: float32 fast family
  int64_t n_bins = 1;
  for (int d = 0; d < n_inputs; ++d) {
    const int64_t nb = n_edges[d] - 1;
    if (nb > 0 && n_bins > (int64_t)1 << 40) {
      delete p;
      return fail(XHIST_ERR_UNSUPPORTED, "histogram has more than 2^40 bins");
    }
    n_bins *= nb;
  }
  p->n_bins = n_bins;
  // unsigned int64-domain inputs: flipping the sign bit of edges (here) and samples (load_dom) maps
  // the uint64 order onto the int64 order every kernel compares in
  p->uns = uns;
  std::vector<std::vector<uint64_t>> biased(n_inputs);
  const void* eff[XHIST_MAX_DIMS];
  for (int d = 0; d < n_inputs; ++d) {
    eff[d] = edges[d];
    if (uns && dim_dom[d] == 1) {
      const uint64_t* e = static_cast<const uint64_t*>(edges[d]);
      biased[d].resize((size_t)n_edges[d]);
      for (int64_t j = 0; j < n_edges[d]; ++j) biased[d][(size_t)j] = e[j] ^ 0x8000000000000000ull;
      eff[d] = biased[d].data();
    }
  }
  edges = eff;  // from here on: the arrays in the domain the kernels compare in
  std::vector<std::vector<uint64_t>> words(n_inputs);
  std::vector<double> lo(n_inputs), hi(n_inputs);
  for (int d = 0; d < n_inputs; ++d) {
    const int E = (int)n_edges[d];
    words[d].assign((size_t)E, 0);
    memcpy(words[d].data(), edges[d], (size_t)E * 8);
  }
  // every edge array is followed by 4 sentinels that compare false against any sample (NaN), so
  // the linear in-bucket count may read up to 4 entries past a bucket's start unconditionally
  const uint64_t kNaN64 = 0x7ff8000000000000ull;
  for (int d = 0; d < n_inputs; ++d)
    for (int k = 0; k < 4; ++k) words[d].push_back(dim_dom[d] == 0 ? kNaN64 : 0x7fffffffffffffffull);
  // more than 65535 edges in some dimension: only the native set, without bucket tables (the vector
  // family then runs table-free on arithmetic edges, everything else takes the generic family)
  p->huge = max_e > 65535;
  const bool vector_sets = cmp_domain == XHIST_CMP_F64 && !p->huge;
  int rc = build_domain(p, cmp_domain == XHIST_CMP_F64 ? 0 : 1, false, n_inputs, n_edges, words, edges, &p->ts[0][0],
                        mixed ? dim_dom : nullptr);
  if (rc == XHIST_OK && vector_sets) rc = build_domain(p, 0, true, n_inputs, n_edges, words, edges, &p->ts[0][1]);
  const std::vector<std::vector<uint64_t>> words64 = words;  // (float64 edges + sentinels: the packed-entry sets below are built from them)
  if (rc == XHIST_OK && vector_sets) {
    // float32 thresholds: thr_j = smallest float32 >= e_j (then (double)x >= e_j <=> x >= thr_j)
    for (int d = 0; d < n_inputs; ++d) {
      const int E = (int)n_edges[d];
      const double* e = static_cast<const double*>(edges[d]);
      std::vector<float> thr((size_t)E + 6, std::nanf(""));  // >= 4 NaN sentinels after the thresholds
      for (int j = 0; j < E; ++j) {
        float f = (float)e[j];
        if ((double)f < e[j]) f = std::nextafterf(f, INFINITY);
        thr[(size_t)j] = f;
      }
      words[d].assign(((size_t)E + 5) / 2, 0);
      memcpy(words[d].data(), thr.data(), words[d].size() * 8);
    }
    rc = build_domain(p, 2, false, n_inputs, n_edges, words, edges, &p->ts[1][0]);
    if (rc == XHIST_OK) rc = build_domain(p, 2, true, n_inputs, n_edges, words, edges, &p->ts[1][1]);
  }
  if (rc != XHIST_OK) {
    for (auto& dom : p->ts)
      for (auto& t : dom)
        if (t.blob) (void)hipFree(t.blob);
    if (p->ts_pk.blob) (void)hipFree(p->ts_pk.blob);
    if (p->ts_pk32.blob) (void)hipFree(p->ts_pk32.blob);
    delete p;
    return rc;
  }
  if (uns)
    for (int d = 0; d < n_inputs; ++d)
      if (dim_dom[d] == 1) p->ts[0][0].dim[d].xor_bias = (int64_t)0x8000000000000000ull;
  // ---- arithmetic edges: e_j == fl(fl(j * step) + e_0) for every j < nb, step = (e_nb - e_0) / nb ----
  // (what numpy.linspace / histogram_bin_edges produce for `bins=int`).  Checked edge by edge with the
  // two roundings kept apart (volatile product: no fma contraction), and only when bins are well
  // resolved (step >= 4 ulp of the largest magnitude) — the bound count_le_arith's guess relies on.
  if (cmp_domain == XHIST_CMP_F64) {
    bool all = true, all32 = true;
    for (int d = 0; d < n_inputs && all; ++d) {
      const double* e = static_cast<const double*>(edges[d]);
      const int nb = (int)n_edges[d] - 1;
      bool ok = nb >= 1 && std::isfinite(e[0]) && std::isfinite(e[nb]);
      double step = 0.0;
      if (ok) {
        step = (e[nb] - e[0]) / (double)nb;
        const double mag = std::max(std::max(std::fabs(e[0]), std::fabs(e[nb])), e[nb] - e[0]);
        const double ulp = std::nextafter(mag, INFINITY) - mag;
        ok = std::isfinite(step) && step > 0.0 && step >= 4.0 * ulp && std::isfinite(1.0 / step);
      }
      for (int j = 0; j < nb && ok; ++j) {
        volatile double m = (double)j * step;
        ok = (m + e[0]) == e[j];
      }
      if (ok) ok = e[nb] >= e[nb - 1];
      all = ok;
      // delta of bin_arith_fast: how far fl(fl(e_j - e_0) * inv_step) is from j, over EVERY edge (the last included),
      // with the kernel's two operations (no fma can form: the subtraction feeds the product); doubled plus 2^-40 for
      // strictness.  Bins resolved so badly that delta reaches 2^-10 keep the exact compares for every sample.
      double arith_h = 0.0;
      if (ok) {
        const double inv = 1.0 / step;
        double delta = 0.0;
        for (int j = 0; j <= nb; ++j) {
          volatile double off = e[j] - e[0];
          volatile double tj = off * inv;
          delta = std::max(delta, std::fabs(tj - (double)j));
        }
        delta = 2.0 * delta + 0x1p-40;
        if (delta < 0x1p-10) arith_h = 0.5 - delta;
      }
      // float32 samples on these edges, decided in float32 arithmetic (bin_arith32_fast): delta32 over every float32 bin
      // boundary B_j and its float32 predecessor, measured with fmaf — one rounding, the device's v_fma_f32.  Offered when
      // the boundaries are distinct (bins at least a few float32 ulps wide) and delta32 stays below 1/8.
      float a32_scale = 0.0f, a32_bias = 0.0f, a32_h = 0.0f;
      if (ok && std::fabs(e[0]) < 3.0e38 && std::fabs(e[nb]) < 3.0e38) {
        const double inv = 1.0 / step;
        const float sc = (float)inv, bi = (float)(-e[0] * inv);
        bool ok32 = std::isfinite(sc) && sc > 0.0f && std::isfinite(bi) && nb < (1 << 22);
        double delta = 0.0;
        float prev = -INFINITY;
        for (int j = 0; j <= nb && ok32; ++j) {
          float b = (float)e[j];  // boundary: the smallest float32 >= e_j; for the last edge the smallest float32 > e_last
          if (j < nb ? (double)b < e[j] : (double)b <= e[j]) b = std::nextafterf(b, INFINITY);
          ok32 = std::isfinite(b) && b > prev;
          prev = b;
          const float t1 = std::fmaf(b, sc, bi), t0 = std::fmaf(std::nextafterf(b, -INFINITY), sc, bi);
          delta = std::max(delta, std::max(std::fabs((double)t1 - (double)j), std::fabs((double)t0 - (double)j)));
        }
        delta = 2.0 * delta + 0x1p-18;  // doubled, plus the rounding of f = t - floor(t) and of f - 0.5 in float32
        if (ok32 && delta < 0.125) {
          a32_scale = sc;
          a32_bias = bi;
          a32_h = (float)(0.5 - delta);
          if ((double)a32_h > 0.5 - delta) a32_h = std::nextafterf(a32_h, 0.0f);
        }
      }
      if (ok)
        for (auto& dom : p->ts[0]) {
          dom.dim[d].step = step;
          dom.dim[d].inv_step = 1.0 / step;
          dom.dim[d].arith = 1;
          dom.dim[d].arith_h = arith_h;
          dom.dim[d].a32_scale = a32_scale;
          dom.dim[d].a32_bias = a32_bias;
          dom.dim[d].a32_h = a32_h;
          dom.dim[d].a32_top = (float)nb + 0.5f;
        }
      all32 = all32 && ok && a32_h > 0.0f;
    }
    p->arith = all;
    p->arith32 = all && all32;
  }
  // (not for arithmetic edges — bins=int, np.linspace: their digitize is one compare per bucket or table-free, the packed
  //  entries would never be picked, and every plan would pay their construction)
  if (vector_sets && !mixed && !p->arith) {
    int rc = XHIST_OK;
    // packed entries share the LDS with the histogram they serve: whatever the smallest form of this plan's histogram
    // (packed uint16 counters) leaves, at most 32 KiB
    size_t edge_bytes = 0;
    for (int d = 0; d < n_inputs; ++d) edge_bytes += words64[d].size() * 8;
    const size_t hist_min = (((size_t)std::min<int64_t>(n_bins, (int64_t)1 << 24) + 1) / 2 + 32) * 4;
    const size_t fixed = edge_bytes + 16 + 1024 + hist_min;
    // (a histogram that cannot live in LDS at all leaves it to the routing pass of the partitioned mode, whose sort buffers
    //  sit next to the entries: the full 32 KiB)
    const bool beyond_lds = hist_min + edge_bytes + 2048 > p->lds_max;
    size_t budget = beyond_lds ? 32 * 1024 : (p->lds_max > fixed ? p->lds_max - fixed : 0);
    budget = std::min<size_t>(budget, 32 * 1024);  // (C3: 27 KiB are left)
    if (budget >= 16 * 8 * (size_t)n_inputs) rc = build_pack_domain(p, false, n_inputs, n_edges, words64, edges, budget);
    // a histogram that fills most of the LDS leaves too little for a usable grid — but its WEIGHTED form (float64 sums) is
    // beyond LDS anyway and goes to the routing pass, which has room: the full budget then (the LDS kernels check the fit)
    const bool big_hist = hist_min > p->lds_max / 4;
    if (rc == XHIST_OK && !p->pk_np && big_hist && budget < 32 * 1024) rc = build_pack_domain(p, false, n_inputs, n_edges, words64, edges, 32 * 1024);
    // float32 samples: no edges in LDS next to the entries
    const size_t fixed32 = 16 + 1024 + hist_min;
    size_t budget32 = beyond_lds ? 32 * 1024 : std::min<size_t>(p->lds_max > fixed32 ? p->lds_max - fixed32 : 0, 32 * 1024);
    if (rc == XHIST_OK && budget32 >= 16 * 8 * (size_t)n_inputs) rc = build_pack_domain(p, true, n_inputs, n_edges, words64, edges, budget32);
    if (rc == XHIST_OK && !p->pk32_np && big_hist && budget32 < 32 * 1024) rc = build_pack_domain(p, true, n_inputs, n_edges, words64, edges, 32 * 1024);
    if (rc != XHIST_OK) {
      for (auto& dom : p->ts)
        for (auto& t : dom)
          if (t.blob) (void)hipFree(t.blob);
      if (p->ts_pk.blob) (void)hipFree(p->ts_pk.blob);
      if (p->ts_pk32.blob) (void)hipFree(p->ts_pk32.blob);
      delete p;
      return rc;
    }
  }
  *out_plan = p;
  return XHIST_OK;
}


extern "C" int xhist_plan_destroy(xhist_plan* p) {
  if (!p) return XHIST_OK;
  DeviceGuard g;
  if (g.set(p->device) == XHIST_OK) {
    for (auto& dom : p->ts)
      for (auto& t : dom)
        if (t.blob) (void)hipFree(t.blob);
    if (p->ts_pk.blob) (void)hipFree(p->ts_pk.blob);
    if (p->ts_pk32.blob) (void)hipFree(p->ts_pk32.blob);
    for (auto& e : p->ring) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (p->mixed_hint) (void)hipHostFree(p->mixed_hint);
  }
  delete p;
  return XHIST_OK;
}

extern "C" int xhist_plan_set_param(xhist_plan* p, const char* key, int64_t value) {
  if (!p || !key) return fail(XHIST_ERR_INVALID, "plan / key is NULL");
  std::lock_guard<std::mutex> lk(p->mu);
  if (!strcmp(key, "block_threads")) {
    if (value != 0 && (value < 64 || value > 1024 || value % 64)) return fail(XHIST_ERR_INVALID, "block_threads must be a multiple of 64 in [64, 1024]");
    p->block_threads = (int)value;
  } else if (!strcmp(key, "grid_blocks")) {
    if (value < 0) return fail(XHIST_ERR_INVALID, "grid_blocks must be >= 0");
    p->grid_blocks = (int)std::min<int64_t>(value, 1 << 30);
  } else if (!strcmp(key, "force_global")) {
    p->force_global = value != 0;
  } else if (!strcmp(key, "force_generic")) {
    p->force_generic = value != 0;
  } else if (!strcmp(key, "profile_stride")) {
    p->profile_stride = (int)std::max<int64_t>(1, std::min<int64_t>(value, 1 << 20));
    p->n_seen = 0;
  } else if (!strcmp(key, "fused")) {
    p->fused_pref = value < 0 ? -1 : (value > 0 ? 1 : 0);
  } else if (!strcmp(key, "records48")) {
    p->records48_pref = value < 0 ? -1 : 0;
    if (p->mixed_hint) *p->mixed_hint = 0u;  // (setting the knob also forgets what earlier calls saw)
  } else if (!strcmp(key, "exchange")) {
    p->exchange_pref = value < 0 ? -1 : (value > 0 ? 1 : 0);
    p->exchange_skip = 0;
    p->exchange_backoff = 16;
  } else if (!strcmp(key, "exchange_min_pct")) {
    if (value < 0 || value > 100) return fail(XHIST_ERR_INVALID, "exchange_min_pct must be in [0, 100]");
    p->exchange_min_pct = (int)value;
  } else if (!strcmp(key, "exchange_arrive_us")) {
    if (value < 0 || value > 1000000) return fail(XHIST_ERR_INVALID, "exchange_arrive_us must be in [0, 1000000]");
    p->exchange_arrive_us = (int)value;
  } else if (!strcmp(key, "exchange_budget_ms")) {
    if (value < -1 || value > 600000) return fail(XHIST_ERR_INVALID, "exchange_budget_ms must be in [-1, 600000]");
    p->exchange_budget_ms = (int)value;
  } else if (!strcmp(key, "route_spl")) {
    if (value != 0 && value != 4 && value != 8) return fail(XHIST_ERR_INVALID, "route_spl must be 0 (auto), 4 or 8");
    p->route_spl = (int)value;
  } else if (!strcmp(key, "flat_rows")) {
    if (value < -1 || value > 1) return fail(XHIST_ERR_INVALID, "flat_rows must be -1 (off), 0 (auto) or 1 (any row length below 65536)");
    p->flat_rows = (int)value;
  } else if (!strcmp(key, "min_parts")) {
    if (value < 0 || value > 128) return fail(XHIST_ERR_INVALID, "min_parts must be in [0, 128]");
    p->min_parts = (int)value;
  } else if (!strcmp(key, "route_pool_pct")) {
    if (value < 0 || value > 100) return fail(XHIST_ERR_INVALID, "route_pool_pct must be in [0, 100]");
    p->route_pool_pct = (int)value;
    if (value != 0 && p->mixed_hint) p->mixed_hint[1] = 0u;  // (a new setting forgets what earlier calls ran into; 0 keeps the note readable)
  } else if (!strcmp(key, "partition")) {
    p->partition = value > 0 ? 1 : (value < 0 ? -1 : 0);
  } else if (!strcmp(key, "lanes")) {
    p->lanes = value > 0 ? 1 : (value < 0 ? -1 : 0);
  } else if (!strcmp(key, "slices")) {
    p->slices_pref = value > 0 ? 1 : (value < 0 ? -1 : 0);
  } else if (!strcmp(key, "route_grid")) {
    if (value < 0 || value > 4096) return fail(XHIST_ERR_INVALID, "route_grid must be in [0, 4096]");
    p->route_grid = (int)value;
  } else if (!strcmp(key, "acc_grid")) {
    if (value < 0 || value > 4096) return fail(XHIST_ERR_INVALID, "acc_grid must be in [0, 4096]");
    p->acc_grid = (int)value;
  } else if (!strcmp(key, "pack")) {
    p->pack_pref = value > 0 ? 1 : (value < 0 ? -1 : 0);
  } else if (!strcmp(key, "arith")) {
    p->arith_pref = value > 0 ? 1 : (value < 0 ? -1 : 0);
  } else if (!strcmp(key, "arith32")) {
    p->arith32_pref = value > 0 ? 1 : (value < 0 ? -1 : 0);
  } else if (!strcmp(key, "lds_copies")) {
    if (value != 0 && (value < 1 || value > 32 || (value & (value - 1)))) return fail(XHIST_ERR_INVALID, "lds_copies must be a power of two in [1, 32]");
    p->lds_copies = (int)value;
  } else if (!strcmp(key, "profile")) {
    // value = number of most recent executes whose main-kernel duration is kept (0 = off)
    if (value < 0 || value > 4096) return fail(XHIST_ERR_INVALID, "profile must be in [0, 4096]");
    DeviceGuard g;
    if (int rc = g.set(p->device)) return rc;
    while ((int64_t)p->ring.size() < value) {
      hipEvent_t a = nullptr, b = nullptr;
      HIPC(hipEventCreate(&a));
      HIPC(hipEventCreate(&b));
      p->ring.emplace_back(a, b);
    }
    p->profile = (int)value;
    p->n_recorded = 0;
  } else {
    return fail(XHIST_ERR_INVALID, "unknown parameter '%s'", key);
  }
  return XHIST_OK;
}

extern "C" int xhist_plan_describe(xhist_plan* p, char* buf, size_t cap) {
  if (!p || !buf || !cap) return fail(XHIST_ERR_INVALID, "plan / buf is NULL");
  std::lock_guard<std::mutex> lk(p->mu);
  // (the pool-dry note is written by the GPU: it is current once the caller has synchronised with the launch)
  const std::string d = p->desc + ((p->mixed_hint && p->mixed_hint[1]) ? " pool_dry=1" : "");
  strncpy(buf, d.c_str(), cap - 1);
  buf[cap - 1] = 0;
  return XHIST_OK;
}

extern "C" int xhist_plan_profile_read(xhist_plan* p, float* ms, int cap, int* n_out) {
  if (!p || !ms || !n_out || cap < 0) return fail(XHIST_ERR_INVALID, "plan / ms / n_out is NULL");
  std::lock_guard<std::mutex> lk(p->mu);
  *n_out = 0;
  if (!p->profile || p->n_recorded == 0) return XHIST_OK;
  DeviceGuard g;
  if (int rc = g.set(p->device)) return rc;
  const int64_t kept = std::min<int64_t>(p->n_recorded, p->profile);
  for (int64_t k = p->n_recorded - kept; k < p->n_recorded && *n_out < cap; ++k) {
    auto& e = p->ring[(size_t)(k % p->profile)];
    HIPC(hipEventSynchronize(e.second));
    HIPC(hipEventElapsedTime(&ms[*n_out], e.first, e.second));
    ++*n_out;
  }
  p->n_recorded = 0;
  return XHIST_OK;
}
