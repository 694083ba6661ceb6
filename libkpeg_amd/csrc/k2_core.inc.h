// libkpeg_amd/csrc/k2_core.inc.h -- K2 from the point where a workgroup has its tables and its slice of the bit string in
// LDS, every lane knows what K1 left for its sub-sequence and the scan inside the workgroup is made (k2_scan.inc.h): the
// workgroup's offset, the decode loop, the block bounds, the error report.  Not a header: the text of a function body,
// included by k_write (after its loads and staging) and by k_sync_write (after K1's work on the same workgroup and the wait
// for its predecessors).  Expects in scope: a (WriteArgs), S, COMPACT, S420, T, s_pre, s_prer, s_wred, s_wredr, s_bits,
// K2_S_START (first bit of the chunk of lane threadIdx.x), stuffed, nsub, nseg, n_u, i0, i, valid, nrec_i, x_prev, wsum_g,
// wrec_g, g0, w0, K2_WG (the workgroup's index) (and, stats builds, tw0, tw1, st_steps).

    // ... + the workgroup's offset, counted from the start of the restart segment.  A segment that began in
    // this workgroup re-bases on a neighbour's scan value; the one open at the workgroup's first
    // sub-sequence began in an earlier workgroup: its base is that workgroup's offset + its cnt up to there.
    int4 open_base = make_int4(0, 0, 0, 0);
    if (nseg > 1 && g0.li != 0) {
        const uint32_t first0 = a.sub_base[g0.seg], gf = first0 / OWN, lf = first0 - gf * OWN;
        int4 part = make_int4(0, 0, 0, 0);
        for (uint32_t j = threadIdx.x; j < lf; j += SYNC_WG) part = add4(part, a.cnt[gf * OWN + j]);
        for (int o = 32; o > 0; o >>= 1) {
            part.x += __shfl_down(part.x, o);
            part.y += __shfl_down(part.y, o);
            part.z += __shfl_down(part.z, o);
            part.w += __shfl_down(part.w, o);
        }
        if ((threadIdx.x & 63) == 0) s_wred[threadIdx.x >> 6] = part;
        __syncthreads();
        open_base = a.wsum[gf];
        for (int q = 0; q < SYNC_WG / 64; ++q) open_base = add4(open_base, s_wred[q]);
    }
    __syncthreads();

#if KPEG_SYNC_STATS
    const uint64_t tw2 = __builtin_amdgcn_s_memtime();
#endif
    uint32_t err = 0;
    // the block open at this lane's entry, if it ends here: this lane's share of its bound
    bool head = false;
    float hA = 0.0f;
    int hnnz = 0;
    bool hcorner = true;
    uint32_t hgb = 0;
    int hchroma = 0;
    // this lane's share of the block open at its exit: (A, nnz, flags)
    int4 share = make_int4(0, 0, 0, 0);
    uint32_t tail_gb = 0;
    int tail_chroma = 0;
    constexpr int SH_OPEN = 1, SH_CORNER = 2, SH_STARTED = 4;
    int4 pre = make_int4(0, 0, 0, 0);
    SubGeom g;
    g.seg = g.li = g.pstart = g.pend = 0;
    DecState s;
    s.p = s.c = s.k = s.q = 0;
    uint32_t first = 0;
    if (valid) {
        if (stuffed) {
            g.seg = 0;
            g.li = i;
            g.pstart = K2_S_START[threadIdx.x];
            g.pend = K2_S_START[threadIdx.x + 1];
        } else {
            g = sub_geom<S>(a.seg_off, a.sub_base, nseg, n_u, i);
        }
        if (g.li == 0) {
            s.p = g.pstart;
            s.c = 0;
            s.k = 0;
            s.q = 0;
        } else {
            s = unpack_state(x_prev);
            if (stuffed) s.p = g.pstart + (s.p & ((1u << StuffedGeom<S>::VSHIFT) - 1u));   // virtual position: it lies in this lane's chunk
        }
        // block index and DC predictors at entry, relative to the segment start
        first = stuffed ? 0u : a.sub_base[g.seg];
        if (g.li != 0) {
            const int4 loc = s_pre[threadIdx.x];
            if (nseg == 1) {
                pre = add4(wsum_g, loc);
            } else if (first >= i0) {
                const int4 f = s_pre[first - i0];
                pre = make_int4(loc.x - f.x, loc.y - f.y, loc.z - f.z, loc.w - f.w);
            } else {
                const int4 w = wsum_g;
                pre = make_int4(w.x + loc.x - open_base.x, w.y + loc.y - open_base.y, w.z + loc.z - open_base.z, w.w + loc.w - open_base.w);
            }
        }
    }
    // Everybody has read the scan's values: until the shares go there after the loop, a lane's s_pre slot is its ring of four
    // records (compact stream) -- records leave as one aligned 16-byte store per four instead of four 4-byte stores, a quarter
    // of the store instructions and of the cache-line requests behind them.
    __syncthreads();
    if (valid) {
        const uint32_t seg_mcu0 = a.interval ? g.seg * a.interval : 0;
        const uint32_t seg_mcus = a.interval ? min(a.interval, a.nmcu - seg_mcu0) : a.nmcu;
        const bool gray = !S420 && a.gray != 0;     // one block per MCU, stored as the MCU's luma block (stride 3; compact stream: the records' block numbers and the DC values likewise, the chroma blocks' DC values are zeroed by the host)
        const uint32_t bstride = gray ? 3u : 1u;
        constexpr uint32_t BPM = S420 ? 6u : 3u;       // blocks per MCU (4:2:0, extension: Y Y Y Y Cb Cr)
        const uint32_t blk_limit = gray ? seg_mcus : seg_mcus * BPM;   // blocks of this segment
        uint32_t b = (uint32_t)pre.x;                  // blocks started so far, within the segment

        BitReader br;
        br.init(s_bits, w0, s.p);
        uint32_t p = s.p, k = s.k, tb = state_table(s);
        const uint32_t tb_wrap = S420 ? 12 * LUT_BYTES : (gray ? 2 * LUT_BYTES : 6 * LUT_BYTES);
        // DC predictors rotate with the blocks (4:2:0: one per component, pd0 = Y, pd1 = Cb, pd2 = Cr): pd0 belongs to the next block to start (component b % 3,
        // == the component of the table in use on a valid stream); DCDiff[c] += zz[0] (MCU.cpp:107)
        const uint32_t cb = (gray || S420) ? 0u : b % 3;
        int pd0 = cb == 0 ? pre.y : (cb == 1 ? pre.z : pre.w);
        int pd1 = cb == 0 ? pre.z : (cb == 1 ? pre.w : pre.y);
        int pd2 = cb == 0 ? pre.w : (cb == 1 ? pre.y : pre.z);
        // State of the block in progress.  A lane that enters inside a block carries that block as its HEAD: the first
        // block end it meets closes a block somebody else began (its bound is settled after the loop, from the shares);
        // every later block began here with its DC symbol.
        // (flags as bits of one register: booleans carried round the loop would live in SGPR masks that the compiler
        // re-merges every iteration, a dozen scalar instructions each time)
        constexpr uint32_t FL_INHEAD = 1;    // the block in progress began in an earlier lane
        constexpr uint32_t FL_HADHEAD = 2;   // ... and ended here
        constexpr uint32_t FL_HNC = 4;       // ... with a term outside the 2x2 corner
        constexpr uint32_t FL_LAST = 8;      // the segment's last block ended here
        uint32_t fl = s.k != 0 ? FL_INHEAD : 0u;
        uint32_t keep = s.q ? E_REC : 0u;    // E_REC while the block keeps its AC terms: quirk Q1, a DC "EOB" drops them (MCU.cpp:97-100)
        uint32_t ncw = 0;                    // bit 31: a non-zero AC term outside the 2x2 corner (natural 1, 8, 9 = zig-zag 1, 2, 4)
        uint32_t ebits = 0;                  // E_BAD / E_DCRUN of every entry met
        uint32_t over = 0;                   // entries whose run went past the end of a block (E_ACSYM of them)
        uint32_t dcrange = 0;
        // compact stream: where this lane's records go (K1 counted them: the scan gives every lane its first ordinal), and
        // which block of which K4 tile the next block to start is (global block gbase + b = 24 tile + bmn)
        uint32_t ord = 0, ord_end = 0, bmn = 0, tn = 0, bm_cur = 0;
        uint32_t nq = 0;                     // records in the ring
        uint32_t* const ring = reinterpret_cast<uint32_t*>(&s_pre[threadIdx.x]);
        if (COMPACT) {
            ord = wrec_g + s_prer[threadIdx.x];
            ord_end = min(ord + nrec_i, a.rec_cap);
            const uint32_t gbn = seg_mcu0 * 3 + b * bstride;
            tn = gbn / TILE_BLOCKS;
            bmn = gbn - tn * TILE_BLOCKS;
            bm_cur = bmn ? bmn - bstride : TILE_BLOCKS - bstride;   // the block in progress at entry (if any)
        }
        float Asum = 0.0f;      // K4's error bound for this block: A = sum |in|, nnz = non-zero AC terms (idct_colour.hip.h)
        int nnz = 0;
        const uint32_t gbase = seg_mcu0 * BPM;
        uint32_t gb = gbase + (b ? b - 1 : 0) * bstride;          // block in progress
        constexpr uint32_t CHROMA_TB = S420 ? 8 * LUT_BYTES : 2 * LUT_BYTES;   // the first chroma table slot
        uint32_t cur_chroma = tb >= CHROMA_TB ? 1u : 0u;          // ... and whether it is a chroma block
        // What this lane may touch: a corrupt stream can count more blocks than the segment has.  No block beyond the
        // segment's last is ever started (the loop ends with the block that completes the segment), so only the block in
        // progress at entry can lie outside: such a lane does nothing.
        uint32_t pend = g.pend;
        if (k == 0 && b >= blk_limit) pend = 0;                   // the segment is complete
        if (k != 0 && !(b >= 1 && b - 1 < blk_limit)) {
            if (b == 0) err |= 64;                                // inside a block before the segment's first one began
            pend = 0;                                             // (else: bits after the segment's last block, ignored as the reference ignores them)
            fl = 0;
        }
        const float m00_l = T.zzm[0][0].x, m00_c = T.zzm[1][0].x;
        uint32_t e1 = lut_first(T, phys_table<S420>(tb), br.peek());
        while (p < pend) {
            const uint32_t win = br.peek();
            const uint32_t tbo = tb;
            const uint32_t e = lut_finish(T, phys_table<S420>(tb), win, e1);
#if KPEG_SYNC_STATS
            st_steps++;
#endif
            p += e & 31;
            br.consume(e & 31);
            const uint32_t kraw = k + ((e >> 16) & 127);
            const bool adv = kraw >= 64;
            const bool isdc = (e & E_ISDC) != 0;
            const bool chroma = tb >= CHROMA_TB;                  // the table in use: chroma tables
            k = adv ? ((e >> 14) & 1u) : kraw;
            tb += adv ? LUT_BYTES : 0u;
            tb = tb == tb_wrap ? 0u : tb;
            e1 = lut_first(T, phys_table<S420>(tb), br.peek());   // next symbol's entry on its way
            const int ext = extend_win(win, (e >> 5) & 31, (e >> 10) & 15);
            ebits |= e;
            over |= kraw > 64 ? e : 0u;
            if (isdc) {
                // a DC symbol opens block gbase + b; its value is coefficient 0 (DC predictors: DCDiff[c] += zz[0], MCU.cpp:107)
                int n;
                if (S420) {
                    n = (tbo < 8 * LUT_BYTES ? pd0 : (tbo == 8 * LUT_BYTES ? pd1 : pd2)) + ext;
                    pd0 = tbo < 8 * LUT_BYTES ? n : pd0;
                    pd1 = tbo == 8 * LUT_BYTES ? n : pd1;
                    pd2 = tbo == 10 * LUT_BYTES ? n : pd2;
                } else {
                    n = pd0 + ext;
                    pd0 = gray ? n : pd1;
                    pd1 = gray ? pd1 : pd2;
                    pd2 = gray ? pd2 : n;
                }
                dcrange |= (uint32_t)(n + 32768);   // bits above 15: the absolute DC does not fit the int16 coefficient layout
                gb = gbase + b * bstride;
                b++;
                cur_chroma = chroma ? 1u : 0u;
                keep = (e >> 25) & 1u ? E_REC : 0u; // E_KEEP
                if (COMPACT) {
                    bm_cur = bmn;
                    if (bmn == 0 && tn <= a.ntiles) a.tile_start[tn] = ord;   // this tile's records begin here
                    bmn += bstride;
                    if (bmn == TILE_BLOCKS) {   // (TILE_BLOCKS is a multiple of the stride)
                        bmn = 0;
                        tn++;
                    }
                    a.dc16[gb] = (int16_t)n;
                } else {
                    a.coef[(size_t)gb << 6] = (int16_t)n;
                }
                // == the terms of block_ebound()'s A in idct_colour.hip.h (DC: 0.25 cc00 Q00, any rounding order is inside U's slack)
                Asum = fabsf((float)n * (chroma ? m00_c : m00_l));
                nnz = 0;
                ncw = 0;
            } else if ((e & keep) && kraw <= 64) {
                // a non-zero AC coefficient (category > 0) at zig-zag position kraw - 1 of a block that keeps its AC terms:
                // exactly what K1 counted as a record
                const float2 zm = T.zzm[chroma ? 1 : 0][kraw - 1];   // scale, natural position << 8 | outside-the-corner << 31
                const uint32_t zw = __float_as_uint(zm.y);
                if (COMPACT) {
                    if (ord < ord_end) {   // (always: K1 counted by the same rule)
                        ring[nq] = ((uint32_t)ext << 16) | (zw & 0x3F00u) | bm_cur;
                        nq++;
                        ord++;
                        if ((ord & 3u) == 0) {
                            // a 16-byte boundary of the record array: the ring goes (whole: one store; a lane's first, shorter run: singly)
                            const uint4 v = *reinterpret_cast<const uint4*>(ring);
                            if (nq == 4) {
                                *reinterpret_cast<uint4*>(a.rec + (ord - 4)) = v;
                            } else {
                                uint32_t* d = a.rec + (ord - nq);
                                d[0] = v.x;
                                if (nq > 1) d[1] = v.y;
                                if (nq > 2) d[2] = v.z;
                            }
                            nq = 0;
                        }
                    }
                } else {
                    a.coef[((size_t)gb << 6) | ((zw >> 8) & 63u)] = (int16_t)ext;
                }
                Asum += fabsf((float)ext * zm.x);
                nnz++;
                ncw |= zw;
            }
            if (adv && !isdc) {
                // the block is complete
                if (b >= blk_limit) {
                    pend = 0;                       // ... and with it the segment: nothing after it is this lane's (or anybody's)
                    fl |= FL_LAST;
                }
                if (fl & FL_INHEAD) {
                    fl = (fl & ~FL_INHEAD) | FL_HADHEAD | ((int)ncw < 0 ? FL_HNC : 0u);
                    hA = Asum;
                    hnnz = nnz;
                    hgb = gb;
                    hchroma = (int)cur_chroma;
                } else {
                    // == block_ebound() of idct_colour.hip.h (range guard: +inf sends the whole block to the exact path)
                    const float E = nnz ? (0x1.004p-24f * Asum) * ((float)nnz + 14.5f) : 0.0f;
                    // even mantissa, rounded up; lowest bit = chroma samples may leave the f32 colour range (Asum >= 249)
                    const uint32_t Eb = ((__float_as_uint(E) + 1u) & ~1u) | ((cur_chroma && !(Asum < 249.0f)) ? 1u : 0u);
                    const float Ef = __uint_as_float(Eb);
                    a.ebound[gb] = !(Asum < 2040.0f) ? __builtin_inff() : ((int)ncw < 0 ? Ef : -Ef);
                }
                Asum = 0.0f;
                nnz = 0;
                ncw = 0;
            }
        }
        if (COMPACT && nq) {   // what is left in the ring (fewer than four)
            const uint4 v = *reinterpret_cast<const uint4*>(ring);
            uint32_t* d = a.rec + (ord - nq);
            d[0] = v.x;
            if (nq > 1) d[1] = v.y;
            if (nq > 2) d[2] = v.z;
        }
        if (ebits & E_BAD) err |= 8;
        if (ebits & E_DCRUN) err |= 16;  // DC symbol with a run nibble: outside the contract
        if (over & E_ACSYM) err |= 32;   // run past the end of a block
        // the reference keeps its DC predictors as ints (MCU.cpp:107-112); one that leaves int16 would wrap here
        // silently: outside the contract, reported instead
        if (dcrange >> 16) err |= KPEG_ERR_DC_RANGE;
        head = (fl & FL_HADHEAD) != 0;
        hcorner = !(fl & FL_HNC);
        tail_gb = gb;
        tail_chroma = (int)cur_chroma;
        if (k != 0 && pend != 0)   // (pend == 0: the lane did nothing, or the segment's last block ended here -- then k == 0)
            share = make_int4(__float_as_int(Asum), nnz, SH_OPEN | ((int)ncw < 0 ? 0 : SH_CORNER) | ((fl & FL_INHEAD) ? 0 : SH_STARTED), 0);
        // end of the last tile: by the lane in which the stream's last block ended (bits after it are ignored, as the
        // reference ignores them: a later lane never gets here)
        if (COMPACT && (fl & FL_LAST) && g.seg + 1 == nseg) a.tile_start[a.ntiles] = ord;
        // the last sub-sequence of a segment must have produced the segment's last block, all of it
        if (g.li + 1 == (stuffed ? nsub : a.sub_base[g.seg + 1] - first)) {
            if (b < blk_limit) err |= 128;
            if (k != 0 && b >= 1 && b - 1 < blk_limit) err |= 64;   // (bits after the segment's last block are ignored, as the reference ignores them)
        }
    }
#if KPEG_SYNC_STATS
    const uint64_t tw3 = __builtin_amdgcn_s_memtime();   // this wavefront's own decode loop is over
#endif
    __syncthreads();   // every lane has read its s_pre
    s_pre[threadIdx.x] = share;
    __syncthreads();
    // (A, nnz, all-in-corner) of a split block as one 64-bit word for the exchange slots; never 0
    auto pack = [](float A, int n, bool crn) -> unsigned long long {
        return (unsigned long long)__float_as_uint(A) | ((unsigned long long)(uint32_t)n << 32) | ((unsigned long long)(crn ? 1u : 0u) << 40) |
               (1ull << 63);
    };
    auto settle = [&](uint32_t blk, float A, int n, bool crn, int chroma) {
        const float E = n ? (0x1.004p-24f * A) * ((float)n + 14.5f) : 0.0f;
        const float Ef = __uint_as_float(((__float_as_uint(E) + 1u) & ~1u) | ((chroma && !(A < 249.0f)) ? 1u : 0u));
        // always stored: k_sync_write makes no presets, so a bound left alone would be whatever an earlier call wrote there
        a.ebound[blk] = A < 2040.0f ? (crn ? -Ef : Ef) : __builtin_inff();
    };
    const uint32_t nown = min((uint32_t)OWN, nsub - i0);
    const bool tail = threadIdx.x == nown - 1 && (share.z & SH_OPEN);   // the workgroup's last lane leaves a block open
    for (int side = 0; side < 2; ++side) {
        // side 0: the block open at entry that ended here; side 1: the block the workgroup's last lane leaves open
        if (side == 0 ? !head : !tail) continue;
        // add the shares of the lanes before this one, back to the lane where the block began
        float A = side == 0 ? hA : __int_as_float(share.x);
        int n = side == 0 ? hnnz : share.y;
        bool crn = side == 0 ? hcorner : (share.z & SH_CORNER) != 0;
        bool found = side == 1 && (share.z & SH_STARTED), broken = false;
        for (int j = (int)threadIdx.x - 1; j >= 0 && !found && !broken; --j) {
            const int4 sh = s_pre[j];
            if (!(sh.z & SH_OPEN)) {
                broken = true;   // inconsistent (corrupt stream): the preset +inf stands
            } else {
                A += __int_as_float(sh.x);
                n += sh.y;
                crn = crn && (sh.z & SH_CORNER);
                found = (sh.z & SH_STARTED) != 0;
            }
        }
        if (side == 0 && found) settle(hgb, A, n, crn, hchroma);
        // the part of the block on this side of a workgroup boundary: swap it for the other side's
        const bool to_prev = side == 0 && !found && !broken && K2_WG > 0;   // began before this workgroup
        const bool to_next = side == 1 && found;                                  // goes on after this workgroup
        if (to_prev || to_next) {
            const unsigned long long other = atomicExch(&a.bslot[to_prev ? K2_WG - 1 : K2_WG], pack(A, n, crn));
            if (other) {
                // the sum runs in stream order on both sides: earlier part + later part
                const float Ao = __uint_as_float((uint32_t)other);
                const int no = (int)((other >> 32) & 0xFF);
                const bool co = ((other >> 40) & 1) != 0;
                settle(to_prev ? hgb : tail_gb, to_prev ? Ao + A : A + Ao, n + no, crn && co, to_prev ? hchroma : tail_chroma);
            }
        }
    }
    if (err) atomicOr(&a.status[1], err);
#if KPEG_SYNC_STATS
    {
        const uint64_t tw4 = __builtin_amdgcn_s_memtime();
        uint32_t mx = st_steps, sm = st_steps;
        for (int o = 32; o > 0; o >>= 1) {
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
            sm += (uint32_t)__shfl_xor((int)sm, o);
        }
        const uint32_t wid = K2_WG * (SYNC_WG / 64) + (threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 0 && wid < 8192) {
            unsigned long long* o = &g_ent_stamp[1][wid * 16];
            o[0] = tw0;
            o[1] = tw1;
            o[2] = tw2;
            o[3] = tw3;
            o[4] = tw4;
            o[5] = __builtin_amdgcn_s_memrealtime();
            o[6] = ((unsigned long long)mx << 32) | sm;
            o[7] = 0;
        }
    }
#endif
