"""Variant sets for experiments/variants.py: name -> [(file, old, new), ...]"""
TC = "tag_continuous.hip"

_FLUSH_OLD = "  for (int q = lane; q < nvec; q += 64) dv[q] = sv[q];"


def _flush(policy):
    if policy == "nt":
        body = ("{ typedef float v4f_ __attribute__((ext_vector_type(4))); "
                "__builtin_nontemporal_store(((const v4f_ *)sv)[q], &((v4f_ *)dv)[q]); }")
    else:
        body = ('{ typedef float v4f_ __attribute__((ext_vector_type(4))); const v4f_ v_ = ((const v4f_ *)sv)[q]; '
                'asm volatile("global_store_dwordx4 %0, %1, off ' + policy +
                '" :: "v"(&dv[q]), "v"(v_) : "memory"); }')
    return [(TC, _FLUSH_OLD, f"  for (int q = lane; q < nvec; q += 64) {body}")]


SETS = {}
SETS_RETIRED_store_policy = {
    "store_policy": {
        "base": [],
        "nt": _flush("nt"),
        "sc1": _flush("sc1"),
        "sc0sc1": _flush("sc0 sc1"),
        "sc1nt": _flush("sc1 nt"),
    },
}

# ---- shader-clock stamps per wavefront at the phase boundaries of the fast path (experiments/phase_profile.py):
# the probes are in the product source (WD_TC_PROBE*, compiled out by default); the variant only switches them on
PROFILE = [(None, "flag", "-DWD_TC_PROBES=1")]
SETS["profile"] = {"prof": PROFILE}

# ---- wave priority by phase: waves in EARLIER phases win VALU arbitration, so laggards catch up and
# the wavefronts of a SIMD finish together (default arbitration is oldest-first: leaders stay leaders
# and the last wave runs alone, latency-bound)
def _prio(p_head, p_a, p_bc, p_gather):
    return [
        (TC, "  const int env0 = a.env_begin + blockIdx.x * epb;\n  TcIn in;",
         f"  const int env0 = a.env_begin + blockIdx.x * epb;\n  __builtin_amdgcn_s_setprio({p_head});\n  TcIn in;"),
        (TC, "  if (active && sg) tc_knn_registers<KMAX>(l.xy + el * N, ag, N, K, nid);",
         f"  __builtin_amdgcn_s_setprio({p_a});\n  if (active && sg) tc_knn_registers<KMAX>(l.xy + el * N, ag, N, K, nid);\n"
         f"  __builtin_amdgcn_s_setprio({p_gather});"),
        (TC, "  // B[k], k = 1..K are the K smallest squared distances to OTHER agents",
         f"  __builtin_amdgcn_s_setprio({p_bc});\n  // B[k], k = 1..K are the K smallest"),
    ]


def _cohort(us):
    return [(TC, "  const int env0 = a.env_begin + blockIdx.x * epb;\n  TcIn in;",
             "  const int env0 = a.env_begin + blockIdx.x * epb;\n"
             "  if (blockIdx.x >= (gridDim.x >> 1)) {\n"
             "    const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();\n"
             f"    while (__builtin_amdgcn_s_memrealtime() - t0_ < {int(us * 100)}ull) __builtin_amdgcn_s_sleep(8);\n"
             "  }\n  TcIn in;")]


SETS_RETIRED_sched = {
    "base": [],
    "prio3210": _prio(3, 2, 1, 0),
    "prio3310": _prio(3, 3, 1, 0),
    "prio0123": _prio(0, 1, 2, 3),
    "prio2210": _prio(2, 2, 1, 0),
    "cohort3": _cohort(3),
    "cohort6": _cohort(6),
}


SETS["flags"] = {
    "base": [],
    "noslp": [(None, "flag", "-fno-slp-vectorize")],
    "O2": [(None, "flag", "-O2")],
    "sched_ilp": [(None, "flag", "-mllvm"), (None, "flag", "-amdgpu-schedule-metric-bias=0")],
}


# ---- round 2, second look at the phase priorities (product = 3, 2, 1, 0)
def _prio2(p_head, p_a, p_bc, p_g):
    return [
        (TC, "  __builtin_amdgcn_s_setprio(3);\n  TcIn in;", f"  __builtin_amdgcn_s_setprio({p_head});\n  TcIn in;"),
        (TC, "  __builtin_amdgcn_s_setprio(2);\n  if (active && sg) tc_knn_registers", f"  __builtin_amdgcn_s_setprio({p_a});\n  if (active && sg) tc_knn_registers"),
        (TC, "  __builtin_amdgcn_s_setprio(1);\n  // B[k], k = 1..K", f"  __builtin_amdgcn_s_setprio({p_bc});\n  // B[k], k = 1..K"),
        (TC, "nid, rank);\n  __builtin_amdgcn_s_setprio(1);", f"nid, rank);\n  __builtin_amdgcn_s_setprio({p_g});"),
    ]


SETS_RETIRED_prio2 = {
    "p3211": [],
    "p3210": _prio2(3, 2, 1, 0),
    "p3321": _prio2(3, 3, 2, 1),
    "p3200": _prio2(3, 2, 0, 0),
    "p3220": _prio2(3, 2, 2, 0),
    "p2210": _prio2(2, 2, 1, 0),
    "p3110": _prio2(3, 1, 1, 0),
}


# ---- code size: the K-specialised entry points hold two copies of the fast path (K == KM folded / runtime K)
SETS_RETIRED_codesize = {
    "base": [],
    "exact_only": [
        (TC, "    else tc_fast_impl<KM, false, false>(a, TcFuse{}, tc_smem, kNumAccelerationActions, kNumTurnActions); \\\n", ""),
        (TC, "    else tc_fast_impl<KM, true, false>(a, fz, tc_smem, kNumAccelerationActions, kNumTurnActions); \\\n", ""),
    ],
    "Os": [(None, "flag", "-Os")],
}


SETS["flags2"] = {
    "base": [],
    "Os": [(None, "flag", "-Os")],
    "Oz": [(None, "flag", "-Oz")],
    "O2": [(None, "flag", "-O2")],
    "unroll8": [(None, "flag", "-mllvm"), (None, "flag", "-unroll-threshold=800")],
}


# ---- full observations (generic kernel): cache policy of the seven 16-byte (dword-aligned) row stores
_FO_OLD = "          for (int c = 0; c < 7; ++c) *(TcF4u *)(row + c * W) = TcF4u{v[c][0], v[c][1], v[c][2], v[c][3]};"


def _fo(policy):
    return [(TC, _FO_OLD,
             "          for (int c = 0; c < 7; ++c) { typedef float v4f_ __attribute__((ext_vector_type(4))); "
             "const v4f_ t_ = {v[c][0], v[c][1], v[c][2], v[c][3]}; "
             'asm volatile("global_store_dwordx4 %0, %1, off ' + policy + '" :: "v"(row + c * W), "v"(t_) : "memory"); }')]


SETS_RETIRED_fullobs_store = {"base": [], "sc1": _fo("sc1"), "nt": _fo("nt"), "sc0sc1": _fo("sc0 sc1")}


# (after the groups were aligned to the 16-byte grid of memory)
_FO2_OLD = "              *(float4 *)(row + c * W + s0) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);"


def _fo2(policy):
    return [(TC, _FO2_OLD,
             "              { typedef float v4f_ __attribute__((ext_vector_type(4))); "
             "const v4f_ t_ = {v[c][0], v[c][1], v[c][2], v[c][3]}; "
             'asm volatile("global_store_dwordx4 %0, %1, off ' + policy + '" :: "v"(row + c * W + s0), "v"(t_) : "memory"); }')]


SETS_RETIRED_fullobs_store2 = {"base": [], "sc1": _fo2("sc1"), "nt": _fo2("nt")}
SETS_RETIRED_fullobs_cmp = {"old": [], "base": [], "nt": _fo2("nt")}  # "old" = round-1 slot-indexed quads (built by hand from git)


# ---- flush policy of the fast path, re-checked on the round-2 final kernel (product: sc1)
_WT_OLD = 'asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(val) : "memory")'


def _wt(policy):
    return [(TC, _WT_OLD, 'asm volatile("global_store_dwordx4 %0, %1, off' + (" " + policy if policy else "") +
             '" ::"v"(ptr), "v"(val) : "memory")')]


SETS["flush_policy2"] = {"sc1": [], "plain": _wt(""), "nt": _wt("nt"), "sc0sc1": _wt("sc0 sc1"), "sc1nt": _wt("sc1 nt")}


SETS["prebuilt_pad"] = {"apart": [], "idsout": []}  # two hand-built code objects (build/variants/{prev,pad}.hsaco): bench only


# ---- one-pass packed-key search inside the tick: wave priority during the search, the exact
# fallback compiled out, the old search (timing only)
_PK_CALL = "    if (!tc_knn_packed<KMAX>(l.xy + el * NP, ag, N, K, nid, rank)) {"
_PK_PRIO2 = "  __builtin_amdgcn_s_setprio(2);\n  if (active && sg) {"
_PK_HALF = ("      nxt = tc_load4(cxy, 4 * g + 4);  // (the last prefetch lands in the padding behind the replica's positions)\n"
            "#pragma unroll\n      for (int u = 0; u < 4; ++u) {\n        const float dx = xi - cur.p[u].x, dy = yi - cur.p[u].y;\n"
            "        const float d2 = dx * dx + dy * dy;\n        WD_TC_INSERT_KEY(d2, 4 * g + u);")
SETS_RETIRED_pk_prio = {
    "pk": [],
    "pk_p1": [(TC, _PK_PRIO2, "  __builtin_amdgcn_s_setprio(1);\n  if (active && sg) {")],
    "pk_p3": [(TC, _PK_PRIO2, "  __builtin_amdgcn_s_setprio(3);\n  if (active && sg) {")],
    "pk_half": [(TC, _PK_HALF, _PK_HALF.replace("      nxt = tc_load4", "      if (g == (ng >> 1)) __builtin_amdgcn_s_setprio(1);\n      nxt = tc_load4", 1))],
    "pk_late1": [(TC, "  __builtin_amdgcn_s_setprio(1);\n  // drop the agent's own entry", "  // drop the agent's own entry")],
}

def _pk_drop(expr, to=1):
    return [(TC, _PK_HALF, _PK_HALF.replace("      nxt = tc_load4", f"      if (g == ({expr})) __builtin_amdgcn_s_setprio({to});\n      nxt = tc_load4", 1))]


SETS_RETIRED_pk_prio2 = {
    "half": _pk_drop("ng >> 1"),
    "quarter": _pk_drop("ng >> 2"),
    "three_q": _pk_drop("(3 * ng) >> 2"),
    "eighth": _pk_drop("ng >> 3"),
    "p3_half": _pk_drop("ng >> 1") + [(TC, _PK_PRIO2, "  __builtin_amdgcn_s_setprio(3);\n  if (active && sg) {")],
    "half_to0": _pk_drop("ng >> 1", 0),
}


# ---- priorities again, after the one-pass search (product: 3 | 2 -> 1 at half of the chain | 1)
_P_START = "  __builtin_amdgcn_s_setprio(3);\n  TcIn in;"
_P_SEARCH = "  __builtin_amdgcn_s_setprio(2);\n  if (active && sg) {"
_P_HALF = "      if (g == (ng >> 1)) __builtin_amdgcn_s_setprio(1);"
_P_AFTER = "  __builtin_amdgcn_s_setprio(1);\n\n  // ------------------------------------------------------------ ids out"
SETS_RETIRED_prio3 = {
    "base": [],
    "start2": [(TC, _P_START, "  __builtin_amdgcn_s_setprio(2);\n  TcIn in;")],
    "search3": [(TC, _P_SEARCH, "  __builtin_amdgcn_s_setprio(3);\n  if (active && sg) {")],
    "after0": [(TC, _P_AFTER, "  __builtin_amdgcn_s_setprio(0);\n\n  // ------------------------------------------------------------ ids out")],
    "half0": [(TC, _P_HALF, "      if (g == (ng >> 1)) __builtin_amdgcn_s_setprio(0);")],
    "half0_after0": [(TC, _P_HALF, "      if (g == (ng >> 1)) __builtin_amdgcn_s_setprio(0);"),
                     (TC, _P_AFTER, "  __builtin_amdgcn_s_setprio(0);\n\n  // ------------------------------------------------------------ ids out")],
}


# ---- phase stamps of the fused policy forward (experiments/mlp_profile.py)
MLP = "policy_mlp.hip"
_MLP_DEFS = '''extern "C" { __device__ unsigned long long *mlp_prof_g = nullptr; }
#define MLP_STAMP(k) do { if ((threadIdx.x & 63) == 0 && mlp_prof_g) mlp_prof_g[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
namespace {

typedef float mlp_v16'''
SETS["mlp_profile"] = {"mlp_prof": [
    (MLP, "namespace {\n\ntypedef float mlp_v16", _MLP_DEFS),
    (MLP, "  // first weight chunk, then this lane's part of its observation row", "  MLP_STAMP(0);\n  // first weight chunk, then"),
    (MLP, "  // ---- layer 1: H1^T = relu(W1 . X^T + b1)", "  MLP_STAMP(1);\n  // ---- layer 1"),
    (MLP, "  mlp_relu<TN1>(acc1);", "  MLP_STAMP(2);\n  mlp_relu<TN1>(acc1);"),
    (MLP, "  mlp_relu<TN2>(acc2);", "  MLP_STAMP(3);\n  mlp_relu<TN2>(acc2);"),
    (MLP, "  // ---- softmax per head over the rows of a column", "  MLP_STAMP(4);\n  // ---- softmax"),
    (MLP, "  constexpr int TS = 65;  // tile stride", "  MLP_STAMP(5);\n  constexpr int TS = 65;  // tile stride"),
    (MLP, "  if (valid && p.values && ((r2 >> 2) & 1) == h && r2 < 32 * TN3) p.values[g] = value;", "  if (valid && p.values && ((r2 >> 2) & 1) == h && r2 < 32 * TN3) p.values[g] = value;\n  MLP_STAMP(6);"),
]}


# ---- optimisation level and priorities once more, on the round-2 final kernel (product: -Os; 3 | 2 -> 1 | 1)
SETS_RETIRED_final_check = {
    "base": [],
    "O3": [(None, "flag", "-O3")],
    "O2": [(None, "flag", "-O2")],
    "p_half0_after0": [(TC, _P_HALF, "      if (g == (ng >> 1)) __builtin_amdgcn_s_setprio(0);"),
                       (TC, _P_AFTER, "  __builtin_amdgcn_s_setprio(0);\n\n  // ------------------------------------------------------------ ids out")],
}


# ---- round 3: what bounds the tick?  "What-if" variants that REMOVE one resource's work (results are
# wrong by construction: timing only) and a start stagger by hardware wave slot.
_WT_DEF = ('#define WD_TC_STORE_WT(ptr, val) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(val) : "memory")')
_NOSTORE = [(TC, _WT_DEF, '#define WD_TC_STORE_WT(ptr, val) asm volatile("" ::"v"(ptr), "v"(val) : "memory")')]
_NOFETCH = [(TC, "  if (FUSED) {\n    // this wavefront's rows of both probability slabs -> LDS",
             "  if (false) {\n    // this wavefront's rows of both probability slabs -> LDS")]
_KNN_CALL = "    if (!tc_knn_packed<KMAX, IDB>(sxy, ag, n_cand, K, nid, rank, in_order)) {"
_NOCHAIN = [(TC, _KNN_CALL,
             "    if (!([&]() {\n#pragma unroll\n      for (int k = 0; k < KMAX; ++k) { const int j = ag + k + 1; nid[k] = j >= N ? j - N : j; }\n"
             "      return true; })()) {")]


def _stagger(ticks_per_slot):
    return [(TC, "  __builtin_amdgcn_s_setprio(3);\n  TcIn in;",
             "  __builtin_amdgcn_s_setprio(3);\n  {\n    unsigned hw_;\n"
             '    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));\n'
             f"    const unsigned long long wait_ = (unsigned long long)((hw_ & 15u) * {ticks_per_slot}u);\n"
             "    const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();\n"
             "    while (__builtin_amdgcn_s_memrealtime() - t0_ < wait_) __builtin_amdgcn_s_sleep(4);\n  }\n  TcIn in;")]


SETS["whatif"] = {
    "base": [],
    "nostore": _NOSTORE,
    "nofetch": _NOFETCH,
    "nochain": _NOCHAIN,
    "nostore_nofetch": _NOSTORE + _NOFETCH,
    "stagger75": _stagger(75),
    "stagger150": _stagger(150),
}


# ---- round 3: geometry known at compile time (the reference injects num_agents / num_envs into its
# kernels through a template header at JIT time, template_env_config.h:19-21): how much of the fused
# tick is index arithmetic that folds away when N, the block size and the head sizes are constants?
_FIXGEOM = [
    (TC, "  const int N = a.N, K = EXACTK ? KMAX : a.K;\n  const int F = 7 * K + 1;\n  const int tid = threadIdx.x, T_ = blockDim.x;\n  const int epb = max(1, T_ / N);",
     "  const int N = 105, K = EXACTK ? KMAX : a.K;\n  const int F = 7 * K + 1;\n  const int tid = threadIdx.x, T_ = 128;\n  const int epb = 1;"),
    (TC, "tc_fast_impl<KM, true, true, 7>(a, fz, tc_smem, kNumAccelerationActions, kNumTurnActions)",
     "tc_fast_impl<KM, true, true, 7>(a, fz, tc_smem, 21, 21)"),
]
_NOFETCH_CLEAN = [(TC, "  if (FUSED) {\n    // this wavefront's rows of both probability slabs -> LDS",
                   "  if (FUSED) { for (int i_ = tid; i_ < epb * N * n_acc; i_ += blockDim.x) { slab_acc[i_] = 1.0f / 21.0f; slab_turn[i_] = 1.0f / 21.0f; } }\n"
                   "  if (false) {\n    // this wavefront's rows of both probability slabs -> LDS")]
SETS["fixgeom"] = {
    "base": [],
    "fixgeom": _FIXGEOM,
    "nofetch_clean": _NOFETCH_CLEAN,
}


# ---- round 3: density threshold of the hybrid row gather (product: sparse form when live rows <= 9/16)
_THR = "  if (n_live_rows * 16 <= wrows * 9) {"
SETS["hybrid_thr"] = {
    "thr9": [],
    "thr7": [(TC, _THR, "  if (n_live_rows * 16 <= wrows * 7) {")],
    "thr11": [(TC, _THR, "  if (n_live_rows * 16 <= wrows * 11) {")],
    "thr0_dense_only": [(TC, _THR, "  if (n_live_rows < 0) {")],
}


# ---- round 4: which of the TagGridWorld rollout changes costs time? (same-box A/B: `variants.py bench gw 2 --workload
# tag_gridworld --ticks-per-launch 200 --steps 100 --warmup 10`)
GW = "tag_gridworld.hip"
_GW_NOCACHE = [(GW, "  uint32_t *const s_cache = (uint32_t *)(s_obs + (size_t)A * F);   // [epb][reset_cache_dwords]",
                "  uint32_t *const s_cache = (uint32_t *)(s_obs + (size_t)A * F);\n  reset_cache_dwords = 0;")]
_GW_PHILOX_EVERY_TICK = [(GW, "        const float u = wd_u01_open_closed(wd_tick_draw((uint32_t)idx, epoch0 + (uint32_t)k, (uint32_t)fz.stream_tag, k0, k1,\n                                                        blk, blk_quad));",
                          "        blk_quad = 0xffffffffu;\n        const float u = wd_u01_open_closed(wd_tick_draw((uint32_t)idx, epoch0 + (uint32_t)k, (uint32_t)fz.stream_tag, k0, k1,\n                                                        blk, blk_quad));")]
_GW_ROW_REBUILD = [(GW, "        if (use_full_observation) {\n          // Only 2 N + 1 of a row's 4 N + 1 values change from tick to tick",
                    "        if (false) {\n          // Only 2 N + 1 of a row's 4 N + 1 values change from tick to tick"),
                   (GW, "        if (!use_full_observation) {\n          s_fx[li] = fx;\n          s_fy[li] = fy;\n        }",
                    "        s_fx[li] = fx;\n        s_fy[li] = fy;")]
SETS["gw"] = {"base": [], "nocache": _GW_NOCACHE, "philox_every_tick": _GW_PHILOX_EVERY_TICK, "row_rebuild": _GW_ROW_REBUILD,
              "all_old": _GW_NOCACHE + _GW_PHILOX_EVERY_TICK + _GW_ROW_REBUILD}


# ---- round 4: are the tracked stores of the move / sampling phases followed by waits for them?  (the waitcnt pass
# protects a store's operand registers until the memory counter drains: the T-tick rollouts lost ~1 us per tick to that)
_TC_UNTRACKED = [
    (TC, "  a.loc_x[gi] = px;\n  a.loc_y[gi] = py;\n  a.speed[gi] = v;\n  a.direction[gi] = dir;\n  a.acceleration[gi] = acc;\n  a.edge_pen_arr[gi] = m.edge_pen;",
     "  wd_store_untracked(a.loc_x + gi, px);\n  wd_store_untracked(a.loc_y + gi, py);\n  wd_store_untracked(a.speed + gi, v);\n"
     "  wd_store_untracked(a.direction + gi, dir);\n  wd_store_untracked(a.acceleration + gi, acc);\n  wd_store_untracked(a.edge_pen_arr + gi, m.edge_pen);"),
    (TC, "    if ((in.cleared != 0) != (sg == 0)) a.obs_rows_cleared[gi] = sg ? 0 : 1;", "    if ((in.cleared != 0) != (sg == 0)) wd_store_untracked(a.obs_rows_cleared + gi, sg ? 0 : 1);"),
    (TC, "      a.timestep[env] = t;\n      tb.tstep[el] = t;\n      tb.tfrac[el] = (float)((double)t / (double)a.T);  // float(t) / episode_length, :474",
     "      wd_store_untracked(a.timestep + env, t);\n      tb.tstep[el] = t;\n      tb.tfrac[el] = (float)((double)t / (double)a.T);  // float(t) / episode_length, :474"),
]
SETS["tc_untracked"] = {"base": [], "untracked": _TC_UNTRACKED}


# ---- round 4: ten blocks per CU instead of eight (LDS <= 16 KB per block: 11 staging rows per wavefront instead of 19).
# At 2000 replicas a CU holds 7.8 blocks anyway; at 8000 / 16000 the blocks run in rounds and a block's lifetime is the
# same 28 us whether its neighbours are in step with it or not (profiles/r04_phase_profile_E16000_t300.txt), so the
# number of resident blocks is what bounds the throughput there.  Run with experiments/occupancy_variant.py (it sets the
# host's staging target to the same value).
SETS["occupancy"] = {"base": [], "lds16k": [(TC, "#define WD_TC_STAGE_TARGET 5400", "#define WD_TC_STAGE_TARGET 3300")]}


# ---- round 4: the prefiltered search of replicas of more than 128 agents on / off ("nopre" = the full chain, as before)
SETS["prefilter_big"] = {"base": [], "pop2": [(TC, "constexpr int POPS = (IDB == 10) ? 1 : 2;", "constexpr int POPS = 2;")], "nopre": [(TC, "constexpr bool PRE = (IDB != 7) && (KMAX <= 12);", "constexpr bool PRE = false;")]}


# ---- round 4 (prepared, NOT measured: the GPU budget of the round was spent): the next steps for the prefiltered search of big
# replicas that experiments/offline/knn_prefilter_big_sim.py prices.  "rank_radius": the radius from the (K + 1)-th SMALLEST
# current squared distance to the remembered agents (x 1.1; it provably holds K agents) instead of 1.15 x (K + 3) / n x the
# largest, falling back to that heuristic when fewer than K + 1 remembered agents are still in the game -- 134 -> ~100 chain
# insertions per wavefront at 1005 agents in the replay.  Build + bench: variants.py build / bench prefilter_next --num-runners 1000
_RANK_RADIUS = [
    (TC, """  float far = 0.0f;
  unsigned n = 0u;
#define WD_TC_PREV_DIST(k)                                                                                  \\
  if (k < M) {                                                                                              \\
    const float dx = xi - p##k.x, dy = yi - p##k.y;                                                         \\
    const float d2 = dx * dx + dy * dy;                                                                     \\
    far = fmaxf(far, d2); /* (maxnum: a NaN operand is ignored) */                                          \\
    asm("v_cmp_o_f32 vcc, %1, %1\\n\\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(n) : "v"(d2) : "vcc");       \\
  }""",
     """  float far = 0.0f;
  unsigned n = 0u;
  float dd[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) dd[k] = __builtin_inff();
#define WD_TC_PREV_DIST(k)                                                                                  \\
  if (k < M) {                                                                                              \\
    const float dx = xi - p##k.x, dy = yi - p##k.y;                                                         \\
    const float d2 = dx * dx + dy * dy;                                                                     \\
    far = fmaxf(far, d2); /* (maxnum: a NaN operand is ignored) */                                          \\
    asm("v_cmp_o_f32 vcc, %1, %1\\n\\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(n) : "v"(d2) : "vcc");       \\
    dd[k] = (d2 == d2) ? d2 : __builtin_inff();                                                             \\
  }"""),
    (TC, """  const float T = far * (1.15f * (float)(K + 3)) * __builtin_amdgcn_rcpf((float)n);
  return (n >= 5u) ? __float_as_uint(T) : 0x7f800000u;""",
     """  // the (K + 1)-th smallest of the M current distances (rank by counting, ties by slot: 78 compares once per tick)
  int cnt[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) cnt[k] = 0;
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = i + 1; j < M; ++j) {
      const int c = (dd[j] < dd[i]) ? 1 : 0;
      cnt[i] += c;
      cnt[j] += 1 - c;
    }
  float kth = 0.0f;
#pragma unroll
  for (int i = 0; i < M; ++i) kth = (cnt[i] == K) ? dd[i] : kth;
  const float T = far * (1.15f * (float)(K + 3)) * __builtin_amdgcn_rcpf((float)n);
  return (n >= (unsigned)(K + 1)) ? __float_as_uint(kth * 1.1f) : (n >= 5u) ? __float_as_uint(T) : 0x7f800000u;"""),
]
SETS["prefilter_next"] = {"base": [], "rank_radius": _RANK_RADIUS}


# ---- round 4 (prepared, not measured): how much of the headline tick is index arithmetic on runtime sizes?  The sizes of
# the BASELINE shape as compile-time constants in EVERY entry (bench the headline shape only: other shapes break) -- the upper
# bound of what `_N105`-style specialised entries could give (the TagGridWorld rollout lost half its instructions this way)
SETS["constfold"] = {"base": [], "n105": [
    (TC, "a.done = done_arr; a.timestep = env_timestep_arr; a.N = kNumAgents; a.T = kEpisodeLength;",
     "a.done = done_arr; a.timestep = env_timestep_arr; a.N = 105; a.T = 500;"),
    (TC, "a.turn_actions = turn_actions_arr; a.max_speed = kMaxSpeed; a.K = kNumOtherAgentsObserved;",
     "a.turn_actions = turn_actions_arr; a.max_speed = kMaxSpeed; a.K = 10;"),
    (TC, "tc_smem, kNumAccelerationActions, kNumTurnActions)", "tc_smem, 20, 20)"),
]}
