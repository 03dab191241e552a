// ptm_lanes_kernel.hpp -- the sweep kernel of SMALL populations: one LANE PER DIMENSION of a chain.
//
// The general kernel (ptm_kernels.hpp) gives every chain one lane; when a wave's 64 chains do not share a rung (fewer
// than 64 walkers per rung: the reference's own shape, one ladder of 1024 rungs) there are only a few waves in the
// whole launch and each lane walks through a chain's ~4000 dependent f64 operations: 80 us per sweep of 1024 chains,
// all of it latency.  Here a chain's DP dimensions sit on DP adjacent lanes (2 chains per wave at DP = 32, 4 at 16, ...; at
// DP = 64 -- 33..64 dimensions, which no other sweep kernel is built for -- a wave is one chain; at DP = 128, 65..128
// dimensions, a wave is one chain still and every lane carries TWO dimensions, lane and lane + 64):
// lane d draws normal d, accumulates row d of factor . z and row d of the precision matrix, and the two reductions
// (box test, y.s) cross the chain's lanes through LDS.  The arithmetic of every number is the one of the other kernels
// (same column order, same fma chains, same four interleaved partial sums), so the chains stay bit-identical.
//
// GEN = false: the plain workload (open bounds, all-uniform prior, zero mean, no one-dimensional moves, no mixture, fixed
// ladder); GEN = true: every state-space / prior / proposal flavour of the general kernel -- boundaries and prior factors
// are per-dimension work and sit naturally on the dimension's lane.  The host-callback likelihood keeps the general
// kernel.  History and MAP tracking are carried.
#pragma once
#include "ptm_kernels.hpp"

namespace ptm {

// LDS of the lanes kernels, in doubles: Box-Muller tables | packed precision matrix | per-wave scratch
template <int DP>
constexpr int lanes_p2_doubles() { return DP > 128 ? 0 : ((DP * (DP + 1) / 2 + 1) & ~1); }   // (beyond 128 dimensions the matrix stays in memory: 263 KB at 256)
template <int DP>
constexpr int lanes_lds_doubles(int waves) { return BM_TABLE_DOUBLES + lanes_p2_doubles<DP>() + waves * (3 * (DP > 64 ? DP : 64) + 4 * (DP > 64 ? 1 : 64 / DP)); }

// the block's tables into LDS (all threads of the block; ends with a barrier)
template <int DP>
__device__ __forceinline__ void lanes_stage(const Dev& p, double* lds_all) {
  constexpr int NP2 = DP * (DP + 1) / 2;
  double* p2s = lds_all + BM_TABLE_DOUBLES;
  for (int i = threadIdx.x; i < BM_TABLE_DOUBLES / 2; i += blockDim.x) reinterpret_cast<bm_d2*>(lds_all)[i] = reinterpret_cast<const bm_d2*>(BM_TABLE)[i];
  if constexpr (DP <= 128)
    for (int k = threadIdx.x; k < NP2; k += blockDim.x) p2s[k] = p.P2[k];
  __syncthreads();
}

// One MH_chain::step for the chains of this block's waves: wave `wslot0 + (threadIdx.x >> 6)` of the launch works for CPW
// chains; the k-th chain of the launch is chain cbase + k * cstride (k < nslots).  The plain launch walks a contiguous range
// (cbase = c_begin, cstride = 1); the fused small-ladder kernel walks ONE walker's rungs (cbase = walker, cstride = W).
template <int DP, int KIND, bool GEN>
__device__ __forceinline__ void lanes_body(const Dev& p, double* lds_all, const int wslot0, const int cbase, const int cstride, const int nslots,
                                           const uint64_t step) {   // (the step: a parameter of its own, see decide_body)
  static_assert(DP == 4 || DP == 8 || DP == 16 || DP == 32 || DP == 64 || DP == 128 || DP == 256 || DP == 512 || DP == 1024, "lanes kernel: DP 4 .. 1024");
  constexpr int E = DP > 64 ? DP / 64 : 1;  // dimensions per lane: lane's d, d + 64, ...
  constexpr int LPC = DP / E;               // lanes per chain
  constexpr int CPW = 64 / LPC;             // chains per wave
  constexpr int WS = DP > 64 ? DP : 64;     // one scratch array of a wave
  constexpr int NP2 = DP * (DP + 1) / 2;    // packed precision matrix
  double* p2s = lds_all + BM_TABLE_DOUBLES;                     // [NP2 (+pad)]
  double* wsc = p2s + lanes_p2_doubles<DP>() + (threadIdx.x >> 6) * (3 * WS + 4 * CPW);   // this wave's scratch
  double* vbuf = wsc;             // [CPW][DP] z, then y
  double* sbuf = wsc + WS;        // [CPW][DP] s_i
  double* tbuf = wsc + 2 * WS;    // [CPW][DP] unused tail / flags
  double* pbuf = wsc + 3 * WS;    // [CPW][4]  partial sums

  const int lane = threadIdx.x & 63;
  const int d = lane % LPC, g = lane / LPC;   // (E > 1: this lane's dimensions are d + 64 e)
  int slot = (wslot0 + (threadIdx.x >> 6)) * CPW + g;
  const bool live = slot < nslots;
  if (!live) slot = nslots - 1;   // dead lanes shadow the last chain and write nothing
  const int c = cbase + slot * cstride;
  const int rl = c / p.W;
  const int w = c - rl * p.W;
  const int rg = p.r0 + rl;
  const bool lead = d == 0;
  const int pos = row_pos<DP>(d);   // (the identity but for DP = 32; dimension d + 64 e sits at pos + 64 e)
  double* __restrict__ row = p.x + (size_t)c * DP;
  const bool hist_on = rl < p.hist.rungs, map_on = rl < p.map.rungs;
  auto sync_wave = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
  // a flag of the chain's lead lane, for all its lanes
  auto from_lead = [&](int v) { return __builtin_amdgcn_ds_bpermute(4 * (g * LPC), v); };

  // host-callback likelihood (GEN build): mode 1 = propose pass (proposal, validity, prior -> xprop / lprior_new / gate, nothing
  // else changes), mode 2 = accept pass with the host's llike_new; 0 = fused
  const int mode = GEN ? p.mode : 0;
  // host-side proposal (GEN build): the proposed state is in xprop already, with its log-Hastings ratio, type and validity
  const bool hp = GEN && p.host_prop != 0;
  const int tc = p.touch[c];  // > 0: the rung took part in that many exchange attempts => no MH move this step
  if (tc && mode != 1) {
    // one add_state per attempt (chain.cc:1487-1490,1531-1534,1554-1557); the LAST of them saw the row as it is now
    const unsigned int nh0 = p.nhist[c];
    const unsigned int a = nh0 + (unsigned int)tc - 1u;
    if (hist_on && a % (unsigned int)p.add_every_n == 0u) {
      const long long hrow = 1 + (long long)(a / (unsigned int)p.add_every_n);
      const size_t o = hist_slot(p.hist, hrow, c);
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (live) p.hist.x[o * DP + d + 64 * e] = row[d + 64 * e];
      if (live && lead)
        hist_scalars(p.hist, o, hrow, p.ll[c], p.lp[c], p.naccept[c], p.ntries[c], p.last_type[c],
                     (GEN && p.beta_add) ? p.beta_add[c] : p.beta[rg]);
    }
    int mapw = 0;
    if (map_on && lead && live) {
      const double tl = p.ll[c], tp = p.lp[c];
      const double tb = ((GEN && p.beta_add) ? p.beta_add[c] : p.beta[rg]) * tl;   // the temperature the rung had at that add
      mapw = map_try(p.map, c, tp + tb, tl, tp) ? 1 : 0;
    }
    if (map_on) {
      mapw = from_lead(mapw);
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (mapw && live) p.map.x[(size_t)c * DP + d + 64 * e] = row[d + 64 * e];
    }
  }

  // ---- MH_chain::step (chain.cc:966-1022); touched chains run along (their lanes would idle anyway) and write nothing
  const uint32_t stream = (uint32_t)(w + p.w_off) * (uint32_t)p.Nt + (uint32_t)rg;
  const u32x4 o0 = draw_block(p.seed, TAG_MH, stream, step, 0);
  int type = 0, axis = -1, kmix = 0;
  double mix_scale = 1.0;
  if (GEN && !hp) {
    double f = p.onedfrac[rl];
    if (p.mix_K > 0) {   // proposal_distribution_set::draw: one uniform picks the member (a set of one draws nothing)
      const double* mx = p.mix + (size_t)rl * p.mix_K * 3;
      const double xs = p.mix_K > 1 ? u01(o0.v3) : 0.0;
      kmix = p.mix_K - 1;
      for (int k = p.mix_K - 2; k >= 0; --k)
        if (xs < mx[3 * k]) kmix = k;
      // differential evolution that is not ready yet is passed over (proposal_distribution.cc:111), as in the general kernel
      if (p.de_on && mx[3 * kmix + 1] < 0 && kmix + 1 < p.mix_K && !de_ready(p, p.nhist[c])) kmix += 1;
      mix_scale = mx[3 * kmix + 1];
      f = mx[3 * kmix + 2];
    }
    if (p.any_oned && !tc && f > 0 && u01(o0.v1) < f) { axis = (int)(p.D * u01(o0.v2)); type = 1; }
  }
  // normal d: slot d & 3 of Philox block 1 + d / 4 (two Box-Muller pairs per block)
  double zd[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int de = d + 64 * e;
    zd[e] = 0.0;
    if (!hp) {
      const u32x4 o = draw_block(p.seed, TAG_MH, stream, step, (uint32_t)((de >> 2) + 1));
      const bool hi = (de & 2) != 0;
      double z0, z1;
      boxmuller(hi ? o.v2 : o.v0, hi ? o.v3 : o.v1, lds_all, z0, z1);
      zd[e] = (de & 1) ? z1 : z0;
      if (GEN && axis >= 0 && de != axis) zd[e] = 0.0;   // one-dimensional move (proposal_distribution.hh:197-205)
    }
  }
  // -- gaussian_prop::draw (proposal_distribution.hh:194-218): offset = factor * z, row d on lane d
  double off[E];
#pragma unroll
  for (int e = 0; e < E; ++e) off[e] = 0.0;
  if (mode == 2 || hp) {
    // accept pass: the proposal was drawn and stored by the propose pass; host-side proposal: it was drawn by the host
  } else if (KIND == KIND_DIAG) {
#pragma unroll
    for (int e = 0; e < E; ++e) off[e] = p.prop[(size_t)rl * p.prop_stride + d + 64 * e] * zd[e];
  } else if constexpr (E == 1) {
    vbuf[g * DP + d] = zd[0];
    const double* fac = p.prop + (size_t)rl * p.prop_stride + d;   // column-major [col][row]: T[d][j] at j * DP + d
    double tcol[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) tcol[j] = fac[j * DP];
    sync_wave();
    double acc = 0.0;
    if constexpr (DP <= 8) {   // the shared column order: natural up to 8 dimensions ...
#pragma unroll
      for (int j = 0; j < DP; ++j) acc = __builtin_fma(tcol[j], vbuf[g * DP + j], acc);
    } else {                   // ... else halves of 16 columns, inside a half s + 4k with s outer, k inner
#pragma unroll
      for (int h = 0; h < DP / 16; ++h)
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int j = 16 * h + 4 * k + sl;
            acc = __builtin_fma(tcol[j], vbuf[g * DP + j], acc);
          }
    }
    off[0] = acc;
    sync_wave();   // vbuf is reused below
  } else {
    // more than 64 dimensions: the factor's column entries come from memory as they are used (no register image of a row)
#pragma unroll
    for (int e = 0; e < E; ++e) vbuf[d + 64 * e] = zd[e];
    sync_wave();
    const double* fac = p.prop + (size_t)rl * p.prop_stride + d;
    for (int h = 0; h < DP / 16; ++h)
      for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = 16 * h + 4 * k + sl;
          const double zj = vbuf[j];
#pragma unroll
          for (int e = 0; e < E; ++e) off[e] = __builtin_fma(fac[(size_t)j * DP + 64 * e], zj, off[e]);
        }
    sync_wave();
  }
  const double ll = p.ll[c], lp = p.lp[c];
  if (GEN && p.mix_K > 0 && !hp) {
    type = kmix + 10 * type;   // proposal_distribution.cc:117
#pragma unroll
    for (int e = 0; e < E; ++e) off[e] = mix_scale * off[e];     // the member is scale_k times the rung's factor
  }
  if (hp) type = p.htype[c];   // proposal_distribution::type()
  // -- differential_evolution::draw (proposal_distribution.cc:476-592) for the chains whose member is the one with a negative scale:
  //    de_draw of the general kernel with dimension d on lane d.  Every number is made by the same operations: products rounded
  //    before their sums, the three inner products summed in index order (each lane of the chain walks the chain's terms in LDS).
  const bool de_move = GEN && !hp && !tc && p.de_on && mix_scale < 0;
  double de_hast = 0.0;
  double xde[E];
#pragma unroll
  for (int e = 0; e < E; ++e) xde[e] = 0.0;
  if constexpr (GEN) {
    if (p.de_on && mode == 2 && de_move) { de_hast = p.de_hast[c]; type = kmix + 10 * p.de_type[c]; }
    if (p.de_on && mode != 2 && __builtin_amdgcn_ballot_w64(de_move) != 0ull) {
      const int D = p.D;
      const unsigned int nh0 = p.nhist[c];
      const long long saved = 1 + (long long)((nh0 + (unsigned int)p.add_every_n - 1u) / (unsigned int)p.add_every_n);   // rows of the ring so far
      const long long rows = p.de_init_extra + saved;
      auto pick = [&](double u) -> const double* {
        const long long spare = rows - 100ll * D;
        const long long first = (spare * (1 - p.de_ignore) > 10ll * D) ? (long long)(spare * p.de_ignore) : 0;
        const long long r = (long long)(first + (rows - first) * u);
        if (r < p.de_init_extra) return p.de_init + ((size_t)r * p.Nc + c) * DP;
        const long long hr = r - p.de_init_extra;
        if (saved - hr > p.hist.cap) atomicOr(p.err, 64);   // the ring has lost that row
        return p.hist.x + hist_slot(p.hist, hr, c) * DP;
      };
      // (history rows are read past the L1: in a kernel that walks several steps -- ptm_fused_kernel.hpp -- the block wrote the newest
      //  of them itself, and a neighbour chain's row may share the line)
      auto hload = [](const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
      const u32x4 b0 = draw_block(p.seed, TAG_MH, stream, step, 0x0DE00000u);
      const bool snk = de_move && p.de_snooker > u01(b0.v0);
      const double* z1 = row;
      const double* z2 = row;
      if (de_move) { z1 = pick(u01(b0.v2)); z2 = pick(u01(b0.v3)); }
      double xr[E], z1d[E], z2d[E];
#pragma unroll
      for (int e = 0; e < E; ++e) { xr[e] = row[pos + 64 * e]; z1d[e] = hload(z1 + pos + 64 * e); z2d[e] = hload(z2 + pos + 64 * e); }
      if (de_move && !snk) {                               // draw_standard
        const double gamma = u01(b0.v1) < p.de_gamma_one ? 1.0 : p.de_gamma_std;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double t1 = z1d[e] * gamma;
          const double a = xr[e] + t1;
          const double t2 = z2d[e] * (-gamma);
          xde[e] = d + 64 * e < D ? a + t2 : 0.0;
        }
      }
      int dt = 0;
      if (__builtin_amdgcn_ballot_w64(snk) != 0ull) {      // draw_snooker (the whole wave walks along: the LDS hand-overs are the wave's)
        const double gamma = (1.2 + u01(b0.v1)) / p.de_gamma_div;
        // the chain's sum of its lanes' terms, in index order
        auto chain_sum = [&](const double (&term)[E], bool mine, double& out) {
#pragma unroll
          for (int e = 0; e < E; ++e) sbuf[g * DP + d + 64 * e] = term[e];
          sync_wave();
          if (mine) {
            double s = 0.0;
            for (int j = 0; j < D; ++j) s = s + sbuf[g * DP + j];
            out = s;
          }
          sync_wave();
        };
        double zz[E], ax[E], term[E];
#pragma unroll
        for (int e = 0; e < E; ++e) zz[e] = ax[e] = 0.0;
        double axis2 = 0.0;
        bool failed = false;
        for (int tries = 0;; ++tries) {                    // the history repeats states: z must differ from the current state
          bool need = snk && !failed && axis2 == 0.0;
          if (need && tries > 1000) { failed = true; need = false; }   // (the reference exits here; the engine raises an error bit and rejects the move)
          if (__builtin_amdgcn_ballot_w64(need) == 0ull) break;
          if (need) {
            const u32x4 bt = draw_block(p.seed, TAG_MH, stream, step, 0x0DE00001u + (uint32_t)tries);
            const double* z = pick(u01(bt.v0));
#pragma unroll
            for (int e = 0; e < E; ++e) { zz[e] = hload(z + pos + 64 * e); ax[e] = xr[e] + zz[e] * (-1.0); }
          }
#pragma unroll
          for (int e = 0; e < E; ++e) term[e] = ax[e] * ax[e];
          chain_sum(term, need, axis2);
        }
        const bool go = snk && !failed;
        double proj = 0.0, fz2 = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double a = z1d[e] * gamma, b = z2d[e] * (-gamma);
          const double diff = a + b;
          term[e] = diff * ax[e];
        }
        chain_sum(term, go, proj);
        if (go) proj = proj / axis2;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double t = ax[e] * proj;
          const double y = xr[e] + t;
          if (go) xde[e] = d + 64 * e < D ? y : 0.0;
          const double f = y + zz[e] * (-1.0);
          term[e] = f * f;
        }
        chain_sum(term, go, fz2);
        if (go) de_hast = (dlog(fz2) - dlog(axis2)) * (double)(D - 1) / 2.0;
        if (snk && failed) {
          if (lead) atomicOr(p.err, 128);
          de_hast = __builtin_nan("");
#pragma unroll
          for (int e = 0; e < E; ++e) xde[e] = d + 64 * e < D ? xr[e] : 0.0;
        }
        if (snk) dt = 1;
      }
      if (de_move) {
        type = kmix + 10 * dt;     // proposal_distribution.cc:117
        if (mode == 1 && live && lead) { p.de_hast[c] = de_hast; p.de_type[c] = dt; }
      }
    }
  }
  double xn[E];
#pragma unroll
  for (int e = 0; e < E; ++e)
    xn[e] = (mode == 2 || hp) ? p.xprop[(size_t)c * DP + pos + 64 * e] : de_move ? xde[e] : row[pos + 64 * e] + off[e];   // state::add (states.cc:205-214)
  // what the state is worth before enforcing: Q9 for a sum built by state::add (on an enforced zero state); a host-side
  // proposal brings its own validity (state::invalid())
  const bool valid0 = hp ? p.hvalid[c] != 0 : p.origin_valid != 0;
  const double beta = (GEN && p.betaC) ? p.betaC[c] : p.beta[rg];
  const double bl = beta * ll;
  const double cur_lpost = lp + bl;
  const double oldlprior = cur_lpost - bl;  // chain.cc:973
  constexpr unsigned long long GM = LPC == 64 ? ~0ull : ((1ull << (LPC & 63)) - 1ull);
  auto all_of_chain = [&](bool v) { return ((__builtin_amdgcn_ballot_w64(v) >> (g * LPC)) & GM) == GM; };
  bool valid = true;
  double newlprior;
  if (mode == 2) {
    valid = (p.gate[c] & 1) != 0;
    newlprior = p.lprior_new[c];
  } else if (!GEN || p.all_uniform) {
    if (GEN) {
      // stateSpace::enforce (states.cc:86-102), each dimension on its lane; Q9: the sum is built on an enforced zero state
      bool vd = true;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int de = d + 64 * e;
        if (p.has_bounds && de < p.D) vd = boundary_enforce(p.blo[de], p.bhi[de], p.bmin[de], p.bmax[de], xn[e]) && vd;
      }
      valid = valid0 && all_of_chain(vd);
    }
    // the box of the all-uniform prior: every dimension of the chain inside
    bool ind = true;
#pragma unroll
    for (int e = 0; e < E; ++e) ind = ind && !(xn[e] < p.plo[d + 64 * e]) && !(xn[e] > p.phi[d + 64 * e]);
    const bool in = all_of_chain(ind);
    newlprior = (valid && in) ? p.lprior_const : -__builtin_inf();
  } else {
    bool vd = true;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int de = d + 64 * e;
      if (p.has_bounds && de < p.D) vd = boundary_enforce(p.blo[de], p.bhi[de], p.bmin[de], p.bmax[de], xn[e]) && vd;
    }
    valid = valid0 && all_of_chain(vd);
    // mixed_dist_product::evaluate: the factors in four interleaved partial products, combined ((p0 p1) p2) p3
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int de = d + 64 * e;
      sbuf[g * DP + de] = de < p.D ? prior_pdf(p.ptype[de], p.plo[de], p.phi[de], p.pcoef[de], xn[e]) : 1.0;
    }
    sync_wave();
    if (d < 4) {
      double pq = 1.0;
#pragma unroll
      for (int t = 0; t < DP / 4; ++t) pq *= sbuf[g * DP + d + 4 * t];
      pbuf[g * 4 + d] = pq;
    }
    sync_wave();
    const double result = ((pbuf[g * 4 + 0] * pbuf[g * 4 + 1]) * pbuf[g * 4 + 2]) * pbuf[g * 4 + 3];
    sync_wave();   // sbuf / pbuf are reused by the likelihood
    newlprior = valid ? dlog(result) : -__builtin_inf();
  }
  const bool want_like = valid && (newlprior > -1e200 || newlprior - oldlprior > p.min_prior);  // chain.cc:980 (Q1)
  if (mode == 1) {   // hand the proposal to the host, change nothing else
    if (live) {
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (!tc) p.xprop[(size_t)c * DP + pos + 64 * e] = xn[e];
      if (lead) {
        p.lprior_new[c] = newlprior;
        p.gate[c] = tc ? (unsigned char)0 : (unsigned char)((valid ? 1 : 0) | (want_like ? 2 : 0));
      }
    }
    return;
  }
  // -- Gaussian likelihood: s_d = sum_{j<d} 2P_dj y_j + P_dd y_d (one fma chain, j ascending), then y.s in four interleaved
  //    partial sums p_q = sum_{i = q mod 4} y_i s_i (i ascending), combined ((p0 + p1) + p2) + p3
  double quad = 0.0;
  if (mode != 2) {
#pragma unroll
    for (int e = 0; e < E; ++e) vbuf[g * DP + d + 64 * e] = (GEN && p.has_mean) ? xn[e] - p.mean[d + 64 * e] : xn[e];
    sync_wave();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int de = d + 64 * e;
      const double* prow = (DP > 128 ? p.P2 : p2s) + (size_t)de * (de + 1) / 2;
      const double* y = vbuf + g * DP;
      double s = 0.0;
      for (int j = 0; j <= de; ++j) s = __builtin_fma(prow[j], y[j], s);
      sbuf[g * DP + de] = s;
    }
    sync_wave();
    if (d < 4) {
      double pq = 0.0;
  #pragma unroll
      for (int t = 0; t < DP / 4; ++t) pq = __builtin_fma(vbuf[g * DP + d + 4 * t], sbuf[g * DP + d + 4 * t], pq);
      pbuf[g * 4 + d] = pq;
    }
    sync_wave();
    quad = ((pbuf[g * 4 + 0] + pbuf[g * 4 + 1]) + pbuf[g * 4 + 2]) + pbuf[g * 4 + 3];
  }
  double newlike = mode == 2 ? p.llike_new[c] : p.like0 - 0.5 * quad;
  double newlpost = newlike * beta + newlprior;
  if (!want_like) newlike = newlpost = -__builtin_inf();
  double logH = newlpost - cur_lpost;  // gaussian_prop: log_hastings_ratio() == 0
  bool accept = valid;
  if (hp) {                            // chain.cc:989-994: prop.log_hastings_ratio(), NaN => reject
    const double hast = p.hastings[c];
    if (hast != hast) accept = false;
    logH = hast + logH;
  }
  if (GEN && de_move) {                // the same for differential evolution's own ratio
    if (de_hast != de_hast) accept = false;
    logH = de_hast + logH;
  }
  if (accept && logH < 0) accept = dlog_u01(o0.v0) < logH;  // chain.cc:998-1001 (NaN stays accepted)
  (void)tbuf;
  if (hp && live && lead) p.acc_out[c] = tc ? (unsigned char)2 : (unsigned char)(accept ? 1 : 0);

  const bool act = live && !tc;
  int mapw = 0;
  if (act) {
    const int ntries1 = p.ntries[c] + 1;
    const unsigned int nh0 = p.nhist[c];
    const int nacc0 = p.naccept[c];
    if (hist_on && nh0 % (unsigned int)p.add_every_n == 0u) {   // add_state saves this one (chain.cc:935-946)
      const long long hrow = 1 + (long long)(nh0 / (unsigned int)p.add_every_n);
      const size_t o = hist_slot(p.hist, hrow, c);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if (accept) p.hist.x[o * DP + pos + 64 * e] = xn[e];
        else p.hist.x[o * DP + d + 64 * e] = row[d + 64 * e];
      }
      if (lead) {
        if (accept) hist_scalars(p.hist, o, hrow, newlike, newlprior, nacc0 + 1, ntries1, type, beta);
        else hist_scalars(p.hist, o, hrow, ll, lp, nacc0, ntries1, p.last_type[c], beta);
      }
    }
    if (map_on && lead && accept) mapw = map_try(p.map, c, newlpost, newlike, newlprior) ? 1 : 0;   // MAP (chain.cc:931-934)
    // an evolving ladder: the state that stays is added at a NEW temperature and may beat the MAP with it
    else if (GEN && map_on && lead && p.betaC) mapw = map_try(p.map, c, cur_lpost, ll, lp) ? 2 : 0;
  }
  if (map_on) {
    mapw = from_lead(mapw);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (mapw == 1 && act) p.map.x[(size_t)c * DP + pos + 64 * e] = xn[e];
      if (GEN && mapw == 2 && act) p.map.x[(size_t)c * DP + d + 64 * e] = row[d + 64 * e];
    }
  }
  if (act) {
    // (every lane of the chain has read the counters above before the lead lane rewrites them: same wave, program order)
    if (lead) {
      p.ntries[c] += 1;
      p.nhist[c] += 1u;
    }
    if (accept) {
#pragma unroll
      for (int e = 0; e < E; ++e) row[pos + 64 * e] = xn[e];
      if (lead) {
        p.naccept[c] += 1;
        p.last_type[c] = type;
        p.ll[c] = newlike;
        p.lp[c] = newlprior;
      }
    }
  } else if (live && tc && lead) {
    p.nhist[c] += (unsigned int)tc;
    p.touch[c] = 0;
  }
}

template <int DP, int KIND, bool GEN>
__global__ __launch_bounds__(256) void sweep_lanes_kernel(const Dev p) {
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  lanes_stage<DP>(p, lds_all);
  lanes_body<DP, KIND, GEN>(p, lds_all, blockIdx.x * 4, p.c_begin, 1, p.c_end - p.c_begin, p.step);
}

}  // namespace ptm
