/*
 * One segment (glz_encode_tmpl.c:268-491) for the kernel of glz_lone.hip: a GPU
 * that holds fewer sessions than it has CUs gives every session a CU of its own,
 * and what then counts is how soon ONE image is through.  The phases of
 * glz_parse_phases.h run side by side on three windows instead of one after the
 * other on one; nine wavefronts:
 *
 *   wavefront 0       walks window w (glz_window_resolve), and does nothing else
 *   wavefronts 1-4    meanwhile scan window w+1 -- on the guess that the walk leaves
 *                     window w into the 256 positions behind it -- without knowing
 *                     which positions of window w will have been stored to the table
 *   wavefronts 5-8    meanwhile write window w-1's tokens (glz_window_emit)
 *
 * and when the walk is through: wavefronts 5-8 make window w's table stores, and
 * wavefronts 1-4 put right what the scan of window w+1 could not know -- a position
 * whose table slot window w stored to gets that store as its table candidate
 * (glz_window_fixup; both windows' pixels are in LDS) -- or, when the walk went
 * somewhere else, throw the scan away and scan the right window in the next step
 * while wavefront 0 waits.  The stops, the table contents every stop sees and the
 * bytes written are those of the one-window kernel; the windows merely start at
 * other positions, which the format does not know of.
 *
 * Included by glz_parse_phases.h when GLZ_LONE is defined.
 */

#define GLZ_LONE_THREADS 576
enum { GLZ_ROLE_WALK = 0, GLZ_ROLE_SCAN = 1, GLZ_ROLE_EMIT = 2 };

#ifdef GLZ_LONE_STAMPS
__device__ unsigned long long glz_lone_stamp_acc[32];
extern "C" void spice_gpu_debug_lone_stamps(unsigned long long *out, int reset)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(glz_lone_stamp_acc), sizeof(unsigned long long) * 32);
    if (reset) {
        unsigned long long z[32] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(glz_lone_stamp_acc), z, sizeof(z));
    }
}
#define LONE_CLOCK() clock64()
#else
#define LONE_CLOCK() 0ll
#endif

/* the tile of the window that starts at ip: positions ip-1 .. ip+WIN+K+2, and ip-2 apart */
template <int MODE>
__device__ __forceinline__ void glz_load_tile(GlzWin &win, const GLZ_GLOBAL uint8_t *base, int ip, int npix, int t)
{
    for (int k = t; k < GLZ_TILE; k += GLZ_WIN) {
        const int pos = ip - 1 + k;
        win.tile[k] = (pos >= 0 && pos < npix) ? glz_load_px<MODE>(base, pos) : 0u;
    }
    if (t == GLZ_WIN - 1) {
        win.tile_m2 = ip >= 2 ? glz_load_px<MODE>(base, ip - 2) : 0u;
    }
}

/* Window `win` (first position ip) was scanned before the walk of the window in front of it,
 * `prev` (first position ip_prev, same segment), had made its table stores.  A position
 * whose slot one of those stores went to sees the newest of them where the scan saw the
 * table: validation and do_match's rules as in glz_window_scan (2) and (4), for a reference
 * in the same segment whose pixels are in prev's tile.  One position per lane. */
template <int MODE>
__device__ __forceinline__ void glz_window_fixup(GlzWin &win, const GlzWin &prev, uint32_t seg_idx, int ip,
                                                 int ip_prev, int npix, int tid, uint32_t newer_h0, uint32_t newer_h1)
{
    typedef GlzTraits<MODE> T;
    const int p = tid, q = ip + p;
    uint32_t A = win.infoA[p];
    if (!(A & GLZ_A_ACT) || (A & GLZ_A_RLE)) {
        return; /* no table look-up here */
    }
    const uint32_t h = win.hash[p];
    if (h == newer_h0 || h == newer_h1) {
        return; /* the scan already gave this position a store that is newer than prev's */
    }
    const uint32_t key = h & (GLZ_KEYS - 1);
    const uint32_t *ins32 = (const uint32_t *)prev.ins;
    /* the row's eight words and the stores' eight words in one round trip: most positions find
     * nothing, and eight dependent reads would be most of what this step costs */
    uint32_t bm[8];
#pragma unroll
    for (int v = 0; v < 8; v++) {
        bm[v] = prev.bitmap[GLZ_BM(key, (uint32_t)v)] & ins32[v];
    }
    int j = -1;
#pragma unroll
    for (int v = 7; v >= 0; v--) {
        uint32_t bits = j < 0 ? bm[v] : 0u;
        while (bits) {
            const int b = 31 - __clz((int)bits);
            bits &= ~(1u << b);
            if (prev.hash[v * 32 + b] == h) {
                j = v * 32 + b;
                bits = 0;
            }
        }
    }
    if (j < 0) {
        return; /* the slot was not stored to: what the scan read stands */
    }
    const int ref = ip_prev + j;
    const int maxn = (npix - 2) - q; /* the reference's own limit lies further on: ref < q */
    const uint32_t pdist = (uint32_t)(q - ref);
    const bool capped = maxn > GLZ_K;
    const int maxk = capped ? GLZ_K : maxn;
    int len = 0;
    bool run = true;
#pragma unroll
    for (int k = 0; k < GLZ_K; k++) {
        run = run && k < maxk && ((win.tile[p + 1 + k] ^ prev.tile[j + 1 + k]) & T::MASK) == 0;
        len += run ? 1 : 0;
    }
    if (len >= T::MIN_REF && len <= T::MAX_REF) {
        /* glz_encode_tmpl.c:243-261 */
        const int need = glz_ref_bytes(0, pdist) / T::ENC_DIV + 1;
        if (len < need) {
            len = 0;
        }
    }
    A &= ~((31u << GLZ_A_GLEN_SHIFT) | GLZ_A_GMAX);
    if (len >= T::MIN_REF) {
        A |= ((uint32_t)len << GLZ_A_GLEN_SHIFT) | (capped && len == GLZ_K ? GLZ_A_GMAX : 0u);
        win.g_ref[p] = (uint32_t)ref;
        win.g_seg[p] = seg_idx;
        win.g_idist[p] = 0;
        win.g_pdist[p] = pdist;
    }
    win.infoA[p] = A;
}

/* The scanning wavefronts, once a window's scan is through: the long matches the walk will
 * want, measured while it is still busy with the window in front.  A table candidate that
 * agreed over all GLZ_K pixels of the scan's compare costs the walk a compare in global memory,
 * round trip after round trip with nothing else to do (glz_resolve_general); here each of
 * the four wavefronts takes one head (see GlzWin) and has the length ready.  The match that
 * takes the walk out of a window is the last one it makes there, so the heads taken are
 * the window's last four.  May run while the window is being walked (a window that was
 * scanned with the walk waiting for it): a record counts once its bit in heads_ready is set. */
template <int MODE>
__device__ __forceinline__ void glz_window_heads(GlzWin &w, const GlzDevJob &job, const GLZ_GLOBAL uint8_t *base,
                                                 int ip, int npix, int gt, uint32_t &tgt)
{
    GlzShared &sh = glz_sh;
    const int lane = gt & 63, p = gt;
    const int wg = (int)glz_uniform((uint32_t)(gt >> 6));
    glz_group_barrier(&sh.arrive[0], tgt, lane); /* every position's scan results are in LDS */
    const uint32_t A = w.infoA[p];
    const bool gmax = (A & GLZ_A_GMAX) != 0;
    const uint32_t seg = w.g_seg[p];
    const int delta = (int)w.g_ref[p] - (ip + p);
    bool goes_on = false;
    if (p > 0 && gmax) {
        goes_on = (w.infoA[p - 1] & GLZ_A_GMAX) && w.g_seg[p - 1] == seg &&
                  (int)w.g_ref[p - 1] - (ip + p - 1) == delta;
    }
    const uint64_t hm = __ballot(gmax && !goes_on);
    if (lane == 0) {
        w.headmask[wg] = hm;
    }
    glz_group_barrier(&sh.arrive[0], tgt, lane);
    /* this wavefront's head: the wg-th from the window's end */
    int pos = -1, skip = wg;
#pragma unroll
    for (int v = 3; v >= 0; v--) {
        uint64_t m = glz_uniform64(w.headmask[v]);
        const int c = __popcll(m);
        if (pos < 0 && skip >= 0 && skip < c) {
            for (int k = 0; k < skip; k++) {
                m &= ~(1ull << (63 - __clzll((long long)m)));
            }
            pos = v * 64 + 63 - __clzll((long long)m);
        }
        skip -= c;
    }
    int end = 0;
    bool gave_up = false;
    uint32_t hseg = 0;
    int hdelta = 0;
    if (pos >= 0) {
        hseg = glz_uniform(w.g_seg[pos]);
        const int ref = (int)glz_uniform(w.g_ref[pos]);
        const GlzDevSeg rs = glz_seg(job, sh, hseg);
        const GLZ_GLOBAL uint8_t *rbase = glz_seg_base(job, hseg, rs);
        const int q = ip + pos;
        int maxn = (npix - 2) - q;
        const int rmax = (int)rs.nunits - ref;
        maxn = rmax < maxn ? rmax : maxn;
        hdelta = ref - q;
        end = q + glz_match_len<MODE, true>(base, q, rbase, ref, GLZ_K, maxn, lane, GLZ_HEAD_ROUNDS, &gave_up);
    }
    if (lane == 0) {
        w.head_pos[wg] = pos;
        w.head_seg[wg] = hseg;
        w.head_delta[wg] = hdelta;
        w.head_end[wg] = end;
        w.head_done[wg] = gave_up ? 0 : 1;
        __threadfence_block();
        atomicOr(&w.heads_ready, 1u << wg);
    }
}

/* All GLZ_LONE_THREADS lanes call this.  from: first position to parse (2 in an image's
 * first segment, whose two leading pixels are already out as literals). */
template <int MODE>
__device__ __forceinline__ void glz_parse_seg(const GlzDevJob &job, uint32_t seg_idx, int from, int tid)
{
    GlzShared &sh = glz_sh;
    const int lane = tid & 63;
    const int wave = (int)glz_uniform((uint32_t)(tid >> 6));
    const int role = wave == 0 ? GLZ_ROLE_WALK : (wave <= 4 ? GLZ_ROLE_SCAN : GLZ_ROLE_EMIT);
    const int gt = role == GLZ_ROLE_SCAN ? tid - GLZ_WAVE : (role == GLZ_ROLE_EMIT ? tid - 5 * GLZ_WAVE : tid);
    const GlzDevSeg cs = glz_seg_uniform(glz_seg(job, sh, seg_idx));
    const GLZ_GLOBAL uint8_t *base = glz_seg_base(job, seg_idx, cs);
    const int npix = (int)cs.nunits;
    const int ip_limit = npix - 6;

    if (tid == 0) {
        sh.arrive[0] = sh.arrive[1] = 0;
        sh.stores_made = 0;
#ifdef GLZ_LONE_STAMPS
        sh.st_fast = 0;
        sh.st_x[0] = sh.st_x[1] = sh.st_x[2] = sh.st_x[3] = 0;
        for (int k = 0; k < 5; k++) {
            sh.st_gen[k] = sh.st_ngen[k] = 0;
        }
#endif
    }
    __syncthreads();
    uint32_t tgt = 0;      /* arrivals this lane's group barrier waits for next */

    int ip = from;         /* where the next window that must be scanned afresh starts */
    uint32_t pend = 0;     /* table stores the last match left for positions ip-2, ip-1 of such a window */
    int cur = 0;           /* slot of the window the walk takes next */
    bool ready = false;    /* ... it is scanned */
    int cur_ip = 0, ent_i = 0, ent_t0 = -1, ent_t1 = -1;
    int prv = -1, prv_ip = 0; /* slot and start of the window whose tokens are not out yet */
    bool heads_due = false;   /* win[cur]'s long matches have not been measured yet */
    /* win[prv]'s table stores are made at the start of the step after its walk, beside the next
     * walk, and the scanning wavefronts wait for them before they look at the table */
    uint32_t stores_target = 0;                            /* what stores_made reaches when they are all made */
    uint32_t prv_h0 = GLZ_NO_HASH, prv_h1 = GLZ_NO_HASH;   /* slots stored to right behind win[prv]'s ... */
    int prv_next = 0;                                      /* ... for positions prv_next-2, prv_next-1 */
#ifdef GLZ_LONE_STAMPS
    long long a_p1 = 0, a_p2 = 0, a_role = 0, n_steps = 0, n_hits = 0, n_bubbles = 0, n_spec = 0, n_jumps = 0,
              n_jump_hits = 0;
#endif

    while (ready || prv >= 0 || ip < ip_limit) {
        const int nxt = cur == GLZ_SLOTS - 1 ? 0 : cur + 1;
#ifdef GLZ_LONE_STAMPS
        const long long c0 = LONE_CLOCK();
#endif
        /* ---- the three phases, side by side */
        if (role == GLZ_ROLE_WALK) {
            if (ready) {
                SPICE_SETPRIO(3);
                glz_window_resolve<MODE>(sh.win[cur], job, seg_idx, base, cur_ip, npix, lane, ent_i, ent_t0, ent_t1);
                SPICE_SETPRIO(0);
            }
        } else if (role == GLZ_ROLE_SCAN) {
            int s_ip = ip;
            bool scan = ip < ip_limit, jump = false;
            if (ready) {
                if (heads_due) {
                    /* the window under the walk was scanned with the walk waiting for it */
                    glz_window_heads<MODE>(sh.win[cur], job, base, cur_ip, npix, gt, tgt);
                    glz_group_barrier(&sh.arrive[0], tgt, lane);
                }
                /* Where will the walk go from win[cur]?  If one of its measured matches leaves the
                 * window, most likely through the first of them to where it ends; else into the
                 * positions right behind the window. */
                const GlzWin &cw = sh.win[cur];
                s_ip = cur_ip + GLZ_WIN;
                int first = GLZ_WIN;
#pragma unroll
                for (int hd = 0; hd < GLZ_HEADS; hd++) {
                    const int hp = (int)glz_uniform((uint32_t)cw.head_pos[hd]);
                    const int he = (int)glz_uniform((uint32_t)cw.head_end[hd]);
                    if (hp >= 0 && hp < first && glz_uniform((uint32_t)cw.head_done[hd]) != 0 && he >= cur_ip + GLZ_WIN) {
                        first = hp;
                        s_ip = he;
                        jump = true;
                    }
                }
                scan = s_ip < ip_limit;
            }
            GlzWin &w = sh.win[ready ? nxt : cur];
            if (gt == 0) {
                w.start_ip = scan ? s_ip : -1;
                w.jump = jump ? 1 : 0;
            }
            if (scan) {
                glz_load_tile<MODE>(w, base, s_ip, npix, gt);
                if (gt == 0) {
                    w.heads_ready = 0;
                }
                glz_group_barrier(&sh.arrive[0], tgt, lane);
                glz_wait_for(&sh.stores_made, stores_target); /* the table is up to date to win[prv]'s end */
                /* The table stores the last match left behind for positions ip-2 and ip-1
                 * (ip-2 first, as the reference orders them) are the newest entries of their
                 * slots for this window.  Scanned afresh, the window knows of them and they
                 * are made here, now that the pixels are in LDS; scanned where a long match
                 * is expected to end, it counts on both and they are made once the walk has
                 * got there; scanned right behind the window in front, it has them inside it. */
                const uint32_t pd = ready ? (jump ? 3u : 0u) : pend;
                uint32_t th0 = GLZ_NO_HASH, th1 = GLZ_NO_HASH;
                if (pd) {
                    typedef GlzTraits<MODE> T;
                    const uint32_t a = w.tile_m2 & T::MASK, b = w.tile[0] & T::MASK;
                    const uint32_t c = w.tile[1] & T::MASK, d = w.tile[2] & T::MASK;
                    th0 = glz_hash3<MODE>(a, b, c);
                    th1 = glz_hash3<MODE>(b, c, d);
                    if (gt == 0 && !ready) {
                        if (pd & 1u) {
                            glz_htab_store(job, th0, seg_idx, (uint32_t)(s_ip - 2));
                        }
                        if (pd & 2u) {
                            glz_htab_store(job, th1, seg_idx, (uint32_t)(s_ip - 1));
                        }
                    }
                }
                uint32_t my_h;
                bool my_hv;
                glz_window_scan<MODE>(w, job, cs, seg_idx, base, s_ip, npix, gt, pd, th0, th1, my_h, my_hv, tgt);
                if (ready) {
                    glz_window_heads<MODE>(w, job, base, s_ip, npix, gt, tgt);
                }
            }
        } else if (role == GLZ_ROLE_EMIT && prv >= 0) {
            const uint32_t my_h = sh.win[prv].hash[gt];
            const bool my_hv = prv_ip + gt + 2 < npix;
            glz_window_emit<MODE>(sh.win[prv], job, seg_idx, prv_ip, gt, my_h, my_hv, tgt, true, prv_h0, prv_h1);
            if (prv_h0 != GLZ_NO_HASH && gt == 0) {
                glz_htab_store(job, prv_h0, seg_idx, (uint32_t)(prv_next - 2));
                glz_htab_store(job, prv_h1, seg_idx, (uint32_t)(prv_next - 1));
            }
            glz_signal(&sh.stores_made, lane);
            glz_window_emit<MODE>(sh.win[prv], job, seg_idx, prv_ip, gt, my_h, my_hv, tgt, false, GLZ_NO_HASH,
                                  GLZ_NO_HASH);
        }
#ifdef GLZ_LONE_STAMPS
        const long long c1 = LONE_CLOCK();
#endif
        __syncthreads();
#ifdef GLZ_LONE_STAMPS
        const long long c2 = LONE_CLOCK();
#endif
        /* ---- what depends on where the walk ended */
        if (ready) {
            const GlzWin &cw = sh.win[cur];
            const int next_ip = (int)glz_uniform((uint32_t)cw.next_ip);
            const int r0 = (int)glz_uniform((uint32_t)cw.tail0), r1 = (int)glz_uniform((uint32_t)cw.tail1);
            const uint32_t cpend = glz_uniform(cw.pend);
            const int nxt_ip = (int)glz_uniform((uint32_t)sh.win[nxt].start_ip); /* -1: nothing was scanned */
            const bool jump = glz_uniform((uint32_t)sh.win[nxt].jump) != 0;
            const bool spec = nxt_ip >= 0;
            const bool hit = spec && next_ip < ip_limit &&
                             (jump ? next_ip == nxt_ip && cpend == 3u
                                   : next_ip >= nxt_ip && next_ip < nxt_ip + GLZ_WIN);
            /* a jump that came true: the stores behind the match (positions next_ip-2, next_ip-1)
             * follow this window's stores; their pixels are in the scanned window's tile */
            uint32_t th0 = GLZ_NO_HASH, th1 = GLZ_NO_HASH;
            if (hit && jump) {
                typedef GlzTraits<MODE> T;
                const GlzWin &nw = sh.win[nxt];
                const uint32_t a = nw.tile_m2 & T::MASK, b = nw.tile[0] & T::MASK;
                const uint32_t c = nw.tile[1] & T::MASK, d = nw.tile[2] & T::MASK;
                th0 = glz_hash3<MODE>(a, b, c);
                th1 = glz_hash3<MODE>(b, c, d);
            }
            if (role == GLZ_ROLE_SCAN && spec) {
                if (hit) {
                    glz_window_fixup<MODE>(sh.win[nxt], cw, seg_idx, nxt_ip, cur_ip, npix, gt, th0, th1);
                } else if (nxt_ip + gt + 2 < npix) {
                    /* the guess was wrong: the slot's bitmap is clean again for its next window */
                    sh.win[nxt].bitmap[GLZ_BM(sh.win[nxt].hash[gt] & (GLZ_KEYS - 1), (uint32_t)gt >> 5)] = 0;
                }
            }
            __syncthreads();
            prv = cur;
            prv_ip = cur_ip;
            prv_h0 = th0;
            prv_h1 = th1;
            prv_next = next_ip;
            stores_target += 4;
            cur = nxt;
            ip = next_ip;
            heads_due = false;
            if (hit) {
                cur_ip = nxt_ip;
                ent_i = next_ip - nxt_ip;
                ent_t0 = (!jump && r0 >= GLZ_WIN) ? r0 - GLZ_WIN : -1;
                ent_t1 = (!jump && r1 >= GLZ_WIN) ? r1 - GLZ_WIN : -1;
                pend = 0;
            } else {
                ready = false;
                pend = cpend;
            }
#ifdef GLZ_LONE_STAMPS
            n_hits += hit ? 1 : 0;
            n_spec += spec ? 1 : 0;
            n_jumps += jump ? 1 : 0;
            n_jump_hits += (jump && hit) ? 1 : 0;
#endif
        } else {
            /* the scan (if any) has filled win[cur]; the tokens of prv are out */
            if (ip < ip_limit) {
                ready = true;
                heads_due = true;
                cur_ip = ip;
                ent_i = 0;
                ent_t0 = ent_t1 = -1;
            }
            prv = -1;
#ifdef GLZ_LONE_STAMPS
            n_bubbles++;
#endif
        }
#ifdef GLZ_LONE_STAMPS
        const long long c3 = LONE_CLOCK();
        a_role += c1 - c0;
        a_p1 += c2 - c0;
        a_p2 += c3 - c2;
        n_steps++;
#endif
    }
#ifdef GLZ_LONE_STAMPS
    if (lane == 0 && (wave == 0 || wave == 1 || wave == 5)) {
        const int r = wave == 0 ? 0 : (wave == 1 ? 1 : 2);
        atomicAdd(&glz_lone_stamp_acc[r], (unsigned long long)a_role); /* own work in the side-by-side part */
        if (wave == 0) {
            atomicAdd(&glz_lone_stamp_acc[3], (unsigned long long)a_p1);
            atomicAdd(&glz_lone_stamp_acc[4], (unsigned long long)a_p2);
            atomicAdd(&glz_lone_stamp_acc[5], (unsigned long long)n_steps);
            atomicAdd(&glz_lone_stamp_acc[6], (unsigned long long)n_hits);
            atomicAdd(&glz_lone_stamp_acc[7], (unsigned long long)n_bubbles);
            atomicAdd(&glz_lone_stamp_acc[8], (unsigned long long)n_spec);
            atomicAdd(&glz_lone_stamp_acc[20], (unsigned long long)n_jumps);
            atomicAdd(&glz_lone_stamp_acc[21], (unsigned long long)n_jump_hits);
            for (int k = 0; k < 4; k++) {
                atomicAdd(&glz_lone_stamp_acc[22 + k], (unsigned long long)sh.st_x[k]);
            }
            atomicAdd(&glz_lone_stamp_acc[9], (unsigned long long)sh.st_fast);
            for (int k = 0; k < 5; k++) {
                atomicAdd(&glz_lone_stamp_acc[10 + k], (unsigned long long)sh.st_gen[k]);
                atomicAdd(&glz_lone_stamp_acc[15 + k], (unsigned long long)sh.st_ngen[k]);
            }
        }
    }
#endif

    /* ---- leftovers as literals, then the run closes; glz_encode_tmpl.c:466-487 */
    if (role == GLZ_ROLE_WALK) {
        /* stores the last window's last match left behind: no tile follows */
        if (pend & 1u) {
            const uint32_t th0 = glz_hash_at<MODE>(base, ip - 2);
            if (lane == 0) {
                glz_htab_store(job, th0, seg_idx, (uint32_t)(ip - 2));
            }
        }
        if (pend & 2u) {
            const uint32_t th1 = glz_hash_at<MODE>(base, ip - 1);
            if (lane == 0) {
                glz_htab_store(job, th1, seg_idx, (uint32_t)(ip - 1));
            }
        }
        glz_run_from_arena<MODE>(job, base, ip, npix - ip, true, lane);
    }
    __syncthreads();
}
