// libkpeg_amd/csrc/k1_wg_body.inc.h -- the work of ONE workgroup of K1 on its sub-sequences: stage, decode, settle inside
// the workgroup, count, store.  Not a header: the text of a function body, included by k_sync_pass and by k_sync_write (K1's
// pass 0 and K2 in one kernel).  Expects in scope: a (SyncArgs), S, COUNT, S420, p, g, t, i0, entry, stuffed, nsub, nseg,
// n_u, Xb_cur, Xb_prev, mute, the macro K1_BODY_BEFORE_ROUNDS (code or nothing) and the __shared__ arrays T, s_lutx, s_wexit, s_wdone, s_edge, s_start, s_n, s_red, s_redn,
// s_bits, STAGE_CAP.

#if KPEG_SYNC_STATS
    const uint64_t tm0 = __builtin_amdgcn_s_memtime();
#endif
    const uint32_t nown = min((uint32_t)OWN, nsub - i0);
    const uint32_t wu = p == 0 ? min(a.warm, i0) : 0u;
    const uint32_t ibase = i0 - wu, nit = wu + nown;
    StuffedStage<S> stg;
    if (stuffed) stg.begin(a.scan, a.scan_len, ibase, nit + StuffedGeom<S>::EXTRA);   // (its loads fly while the tables load)
    load_tables(&T, a.tabs);
    if (!S420) load_tables_x<(S < SUBSEQ_DENSE)>(s_lutx, a.tabs);
    // stage this workgroup's slice of the bit string (its sub-sequences are contiguous in u)
    uint32_t w0 = 0;
    if (stuffed) {
        stg.finish(s_bits, STAGE_CAP, s_start, s_redn, a.scan_len);
    } else {
        w0 = sub_geom<S>(a.seg_off, a.sub_base, nseg, n_u, ibase).pstart >> 5;
        const uint32_t total_words = (n_u + 3) / 4 + 2;
        stage_bits(s_bits, STAGE_CAP, a.u, w0, total_words > w0 ? total_words - w0 : 0u);
    }
    __syncthreads();
    // stuffed mode: states that leave the workgroup carry virtual positions (chunk << VSHIFT | bit inside the chunk); the
    // decode loops run on positions in s_bits.  An item's entry state lies in its own chunk, its exit state in the next one.
    constexpr uint32_t VSHIFT = StuffedGeom<S>::VSHIFT;
    // (this lane's item only: the bounds of its chunk are kept in registers -- pbeg, pend below -- not read again per round)
    uint32_t pbeg = 0, pend = 0;
    auto to_local = [&](uint64_t v, uint32_t tl) -> DecState {
        DecState d = unpack_state(v);
        if (stuffed) d.p = pbeg + (d.p & ((1u << VSHIFT) - 1u));
        return d;
    };
    auto to_virtual = [&](uint64_t x, uint32_t tl) -> uint64_t {
        if (!stuffed) return x;
        const uint32_t pl = (uint32_t)x;
        return (x & 0xFFFFFFFF00000000ull) | (uint64_t)(((ibase + tl + 1) << VSHIFT) | (pl - pend));
    };

    uint32_t ex_iters = 0;   // (stats builds: the steps of the last exit_of)
    auto exit_of = [&](DecState d, uint32_t pe) __attribute__((always_inline)) -> uint64_t {
        if (S420) {
            const RunResult rx = run_count<COUNT, false, S420>(T, s_bits, w0, d, pe, a.gray != 0);
#if KPEG_SYNC_STATS
            ex_iters = rx.iters;
#endif
            return rx.exit_state;
        }
#if KPEG_SYNC_STATS
        return run_exit<(S < SUBSEQ_DENSE)>(T, s_lutx, s_bits, w0, d, pe, a.gray != 0, &ex_iters);
#else
        return run_exit<(S < SUBSEQ_DENSE)>(T, s_lutx, s_bits, w0, d, pe, a.gray != 0);
#endif
    };
    (void)ex_iters;

    // Every wavefront settles its 64 consecutive items on its own: no barrier, no work list.  An item's entry state
    // is its left neighbour's exit state -- one lane over (DPP shift), for lane 0 the last exit state of the
    // wavefront before, handed over through LDS -- and an item decodes again whenever that differs from the state it
    // last decoded from.  A round costs the longest decode among the lanes that take part, not the longest of the
    // workgroup, and nothing else.  A wavefront is done when the one before it is done and none of its lanes wants
    // another decode.
    const uint32_t lane = t & 63, wave = t >> 6;
    const bool have = t < nit;
    bool segfirst = false;      // opens a restart segment: its entry state is known
    bool fixed = true;          // never decodes again (segfirst; item 0 of pass 0: nothing to check its guess against)
    uint64_t used = 0;          // the state this item last decoded from
    bool dirty = false;         // decoded in this launch: its counts are to be made
    RunResult r;
    r.exit_state = 0;
    r.cnt = make_int4(0, 0, 0, 0);
    r.nrec = 0;
#if KPEG_SYNC_STATS
    uint32_t st_runs = 0, st_iters = 0, st_rounds = 0, st_wait = 0;
    r.iters = 0;
    const uint64_t tm1 = __builtin_amdgcn_s_memtime();
#endif
    if (have) {
        SubGeom geo;
        if (stuffed) {
            geo.seg = 0;
            geo.li = ibase + t;
            geo.pstart = s_start[t];
            geo.pend = s_start[t + 1];
        } else {
            geo = sub_geom<S>(a.seg_off, a.sub_base, nseg, n_u, ibase + t);
        }
        pbeg = geo.pstart;
        pend = geo.pend;
        segfirst = geo.li == 0;
        fixed = segfirst || (p == 0 && t == 0);
        if (p == 0) {
            DecState s;
            s.p = geo.pstart;
            s.c = 0;
            s.k = 0;
            s.q = 0;
            {
                DecState sv = s;
                if (stuffed) sv.p = (ibase + t) << VSHIFT;
                used = pack_state(sv);
            }
            r.exit_state = to_virtual(exit_of(s, pend), t);   // exit state only: see below
            dirty = true;
#if KPEG_SYNC_STATS
            st_runs++;
            st_iters += ex_iters;
#endif
        } else {
            r.exit_state = a.X[i0 + t];
            r.cnt = a.cnt[i0 + t];
            if (COUNT) r.nrec = a.nrec[i0 + t];
        }
    }
    if (p != 0) {
        // the states the loaded results were decoded from: the left neighbour's exit state (they converged in an earlier
        // pass); item 0's was this workgroup's assumption
        const uint64_t left = (uint64_t)wave_shr1((uint32_t)r.exit_state) | ((uint64_t)wave_shr1((uint32_t)(r.exit_state >> 32)) << 32);
        used = lane ? left : (t ? (have ? a.X[i0 + t - 1] : 0ull) : a.assumed[g]);
    }
    // every wavefront's last exit state so far, before anybody looks
    if (wave * 64 < nit && lane == min(63u, nit - 1 - wave * 64)) {
        s_wexit[wave] = r.exit_state;
        s_wdone[wave] = 0u;
    }
#if KPEG_SYNC_STATS
    const uint64_t tm2 = __builtin_amdgcn_s_memtime();
#endif
#ifndef KPEG_ABLATE_NOZERO
    // pass 0 clears the coefficient buffer behind the rounds below, which only touch LDS
    // (k_sync_write, a.gen != 0: no presets -- inside one launch no location may be written by two workgroups with plain
    // stores, the XCDs' L2s are not coherent for those; there every entry is written by the workgroup that owns it, or, corrupt
    // stream, keeps what an earlier call left: K4 clamps what it reads there, and the call fails anyway)
    if (p == 0 && a.gen) {
        if (t == 0) {
            atomicExch(&a.bslot[g], 0ull);
            a.done[g] = 0u;
        }
    } else if (p == 0) {
        const uint32_t nwg = (nsub + OWN - 1) / OWN;   // the workgroups that get here
        if (a.tile_start) {
            // compact coefficient stream: nothing to clear but the tiles' first-record table (a tile whose first block a
            // corrupt stream never starts then reads as empty)
            const uint32_t tper = (a.ntiles + 1 + nwg - 1) / nwg;
            const uint32_t t0 = min(a.ntiles + 1, g * tper), t1 = min(a.ntiles + 1, t0 + tper);
            for (uint32_t q = t0 + t; q < t1; q += SYNC_WG) a.tile_start[q] = 0u;
        } else {
            const uint64_t per = (a.coef_n16 + nwg - 1) / nwg;
            const uint64_t b0 = (uint64_t)g * per, b1 = min(a.coef_n16, b0 + per);
            const uint4 z = make_uint4(0, 0, 0, 0);
            for (uint64_t q = b0 + t; q < b1; q += SYNC_WG) a.coef16[q] = z;
        }
        const uint32_t eper = (a.nblocks + nwg - 1) / nwg;
        const uint32_t e0 = min(a.nblocks, g * eper), e1 = min(a.nblocks, e0 + eper);
        // (grayscale: the MCU's chroma blocks stay all zero -- bound -0.0 = exact and corner-only; only luma is decoded)
        for (uint32_t q = e0 + t; q < e1; q += SYNC_WG) a.ebound[q] = (a.gray && q % 3u) ? 0x80000000u : 0x7F800000u;
        if (t == 0) {
            a.bslot[g] = 0ull;
            a.done[g] = 0u;
        }
    }
#endif
    K1_BODY_BEFORE_ROUNDS   // (k_sync_write's second, strict launch waits here -- its bits and tables staged -- for the entry state it is to start from)
    __syncthreads();
#if KPEG_SYNC_STATS
    const uint64_t tmc = __builtin_amdgcn_s_memtime();
    uint64_t tr[6] = {0, 0, 0, 0, 0, 0};
    uint32_t act[6] = {0, 0, 0, 0, 0, 0};
    uint64_t tmb = 0;   // the rounds are over, the counts' decode begins
#endif
    if (wave * 64 < nit) {
        const uint32_t last_lane = min(63u, nit - 1 - wave * 64);
        SpinGuard guard(K1_SPIN_TICKS);
        for (;;) {
            // the wavefront before: done flag first, then its last exit state (written in the opposite order)
            const uint32_t prev_done = wave ? __hip_atomic_load(&s_wdone[wave - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 1u;
            const uint64_t prev_x = wave ? __hip_atomic_load(&s_wexit[wave - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : entry;
            const uint64_t left = (uint64_t)wave_shr1((uint32_t)r.exit_state) | ((uint64_t)wave_shr1((uint32_t)(r.exit_state >> 32)) << 32);
            // (exit states only, here: what the rounds cost is the longest chain of re-decodes, one lane after the other, so
            // the step of these decodes is kept as short as it can be; the counts come after the rounds, see below)
            const uint64_t in = fixed ? used : (lane ? left : prev_x);
            const bool again = have && in != used;
            if (!__ballot(again)) {
                if (prev_done) break;
                if (guard.expired()) {   // (cannot happen: the wavefronts of a workgroup run together; bounded like every wait)
                    if (lane == 0) atomicOr(&a.status[1], KPEG_ERR_TIMEOUT);
                    break;
                }
#if KPEG_SYNC_STATS
                st_wait++;
#endif
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            if (again) {
                r.exit_state = to_virtual(exit_of(to_local(in, t), pend), t);
                used = in;
                dirty = true;
#if KPEG_SYNC_STATS
                st_runs++;
                st_iters += ex_iters;
#endif
            }
#if KPEG_SYNC_STATS
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if ((int)st_rounds == q) {
                    tr[q] = __builtin_amdgcn_s_memtime();
                    act[q] = (uint32_t)__popcll(__ballot(again));
                }
            st_rounds++;
#endif
            if (lane == last_lane) __hip_atomic_store(&s_wexit[wave], r.exit_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (lane == last_lane) {
            __hip_atomic_store(&s_wexit[wave], r.exit_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&s_wdone[wave], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // The entry states stand: now the counts (blocks, DC sums, records), every item that decoded above once more, all
        // lanes of the wavefront together -- one decode with the longer step instead of one per round.
#if KPEG_SYNC_STATS
        tmb = __builtin_amdgcn_s_memtime();
#endif
        if (dirty && t >= wu) {   // (not the warm-up items: theirs are their own workgroup's to make)
            const uint64_t xs = r.exit_state;
            r = run_count<COUNT, true, S420>(T, s_bits, w0, to_local(used, t), pend, a.gray != 0);
            r.exit_state = xs;
#if KPEG_SYNC_STATS
            st_runs++;
            st_iters += r.iters;
#endif
        }
    }
#if KPEG_SYNC_STATS
    {
        const uint64_t tm3 = __builtin_amdgcn_s_memtime();
        uint32_t sr = st_runs, si = st_iters, mx = st_iters;
        for (int o = 32; o > 0; o >>= 1) {
            sr += (uint32_t)__shfl_xor((int)sr, o);
            si += (uint32_t)__shfl_xor((int)si, o);
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
        }
        const uint32_t wid = g * (SYNC_WG / 64) + wave;
        if (lane == 0 && p == 0 && wid < 8192) {
            unsigned long long* o = &g_ent_stamp[0][wid * 16];
            o[0] = tm0;
            o[1] = tm1;
            o[2] = tm2;
            o[3] = tm3;
            o[4] = __builtin_amdgcn_s_memrealtime();
            o[5] = ((unsigned long long)st_rounds << 32) | sr;
            o[6] = ((unsigned long long)mx << 32) | si;
            o[7] = st_wait;
            o[8] = tmc;
            for (int q = 0; q < 6; ++q) o[9 + q] = tr[q] | ((unsigned long long)act[q] << 56);
            o[15] = tmb;
        }
    }
#endif

    int4 tot = make_int4(0, 0, 0, 0);
    uint32_t trec = 0;
    if (have && t >= wu) {
        a.X[i0 + t - wu] = r.exit_state;
        tot = r.cnt;
        a.cnt[i0 + t - wu] = tot;
        if (COUNT) {
            trec = r.nrec;
            a.nrec[i0 + t - wu] = trec;
        }
        if (t == wu) s_edge[0] = segfirst ? X_NONE : used;   // what this workgroup's first own item decoded from
        if (t == nit - 1) s_edge[1] = r.exit_state;
    }
    // per-workgroup totals for the scan
    tot = make_int4(wave_scan_incl(tot.x), wave_scan_incl(tot.y), wave_scan_incl(tot.z), wave_scan_incl(tot.w));
    trec = COUNT ? wave_scan_incl(trec) : 0u;
    if ((t & 63) == 63) {
        s_red[t >> 6] = tot;
        s_redn[t >> 6] = trec;
    }
    __syncthreads();
    if (t == 0) {
        int4 w = s_red[0];
        uint32_t wr = s_redn[0];
        for (int q = 1; q < SYNC_WG / 64; ++q) {
            w = add4(w, s_red[q]);
            wr += s_redn[q];
        }
        const uint64_t last = s_edge[1];
        const bool known = s_edge[0] == X_NONE;   // first own sub-sequence opens a restart segment
        a.wsum[g] = w;
        a.wrec[g] = wr;
        a.assumed[g] = s_edge[0];
        Xb_cur[g] = last;
        if (a.chained && !mute) {
            __threadfence();
            __hip_atomic_store(&a.done[g], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p == 0 && a.gen) {
            // k_sync_write: published -- every word carries its value and the call's number, stored and polled with relaxed
            // atomics: no fence anywhere (an agent-scope release writes this XCD's dirty L2 lines back, an acquire
            // invalidates: per workgroup and per poll that doubled the kernel's time), no order between the words needed,
            // each one says for itself whether it is there.  (From the registers: nothing is read back.)
            const uint32_t vals[PUB_WORDS] = {(uint32_t)w.x, (uint32_t)w.y, (uint32_t)w.z, (uint32_t)w.w, wr,
                                              (uint32_t)last, (uint32_t)(last >> 32), (uint32_t)s_edge[0], (uint32_t)(s_edge[0] >> 32)};
            for (uint32_t q = 0; q < PUB_WORDS; ++q)
                __hip_atomic_store(&a.pub[(size_t)g * PUB_WORDS + q], (unsigned long long)vals[q] | ((unsigned long long)a.gen << 32), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p == 0) {
            // nothing is verified before pass 1
            if (!known) atomicAdd(&a.meta->moved[0], 1u);
        } else if (!a.chained && last != Xb_prev[g] && i0 + nown < nsub) {
            // the last workgroup has no successor: its movement needs no further pass
            atomicAdd(&a.meta->moved[p], 1u);
        }
    }
