// libkpeg_amd/csrc/k2_scan.inc.h -- K2's scan inside the workgroup: (blocks, DC sums, records) before every sub-sequence of the
// workgroup, from what K1 left for them.  Nothing here depends on another workgroup, so k_sync_write runs it BEFORE it waits for
// its predecessors.  Not a header: the text of a function body, included by k_write and by k_sync_write ahead of
// k2_core.inc.h.  Expects in scope: COMPACT, cnt_i, nrec_i, s_pre, s_prer, s_wred, s_wredr.

    // (blocks, DC sums) before every sub-sequence: exclusive scan of cnt inside the workgroup (inside the wavefronts by
    // shuffles, their totals through LDS: one barrier instead of the twenty of a scan that lives in LDS) ...
    {
        const int4 v = cnt_i;
        const uint32_t vr = nrec_i;
        const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int4 inc = make_int4(wave_scan_incl(v.x), wave_scan_incl(v.y), wave_scan_incl(v.z), wave_scan_incl(v.w));
        const uint32_t incr = COMPACT ? wave_scan_incl(vr) : 0u;
        if (lane == 63) {
            s_wred[wave] = inc;
            if (COMPACT) s_wredr[wave] = incr;
        }
        __syncthreads();
        int4 base = make_int4(0, 0, 0, 0);
        uint32_t baser = 0;
        for (uint32_t q = 0; q < SYNC_WG / 64; ++q)
            if (q < wave) {
                base = add4(base, s_wred[q]);
                if (COMPACT) baser += s_wredr[q];
            }
        s_pre[threadIdx.x] = make_int4(base.x + inc.x - v.x, base.y + inc.y - v.y, base.z + inc.z - v.z, base.w + inc.w - v.w);
        if (COMPACT) s_prer[threadIdx.x] = baser + incr - vr;   // exclusive
        __syncthreads();   // (s_wred is used again below)
    }
