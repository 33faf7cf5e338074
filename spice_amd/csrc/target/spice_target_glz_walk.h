/*
 * The GLZ walk's loop over the ordinary stops of a quarter for gfx950, in GCN assembly:
 * a single wavefront issues about one instruction per five cycles, and this loop is the
 * chain every image's parse hangs on, so it is written out instruction by instruction with
 * its operands in scalar registers.  Included by glz_parse_phases.h, which documents the
 * contract (and defines the GLZ_W_* layout of the quarter word the loop reads).
 */
__device__ __forceinline__ void glz_fast_stops(uint32_t W, int &l, uint64_t &ins, uint64_t &stops,
                                               uint64_t &ms, int &last_len)
{
    uint32_t w, len, t;
    uint64_t bit, tm;
    /* the operands below live in scalar registers; every lane holds the same values */
    l = (int)glz_uniform((uint32_t)l);
    last_len = (int)glz_uniform((uint32_t)last_len);
    ins = glz_uniform64(ins);
    stops = glz_uniform64(stops);
    ms = glz_uniform64(ms);
    asm volatile(
        "s_cmp_lt_i32 %[l], 64\n"
        "s_cbranch_scc0 9f\n"
        "1:\n"
        "s_nop 3\n"                                /* l was just written; it selects the lane */
        "v_readlane_b32 %[w], %[W], %[l]\n"
        "s_cmp_ge_u32 %[w], 0x40000000\n"
        "s_cbranch_scc1 3f\n"                      /* GLZ_W_SLOW or GLZ_W_RUN */
        "s_lshr_b32 %[t], %[w], 6\n"
        "s_bitcmp1_b64 %[ins], %[t]\n"             /* bit index = low six bits: d2 */
        "s_cselect_b32 %[t], 17, 22\n"
        "s_bitcmp1_b64 %[ins], %[w]\n"             /* d1 */
        "s_cselect_b32 %[t], 12, %[t]\n"
        "s_lshr_b32 %[len], %[w], %[t]\n"
        "s_and_b32 %[len], %[len], 31\n"
        "s_cmp_gt_u32 %[len], 29\n"
        "s_cbranch_scc1 9f\n"                      /* GLZ_W_LEN_MAXED, GLZ_W_LEN_DEEP */
        "s_lshl_b64 %[bit], 1, %[l]\n"
        "s_or_b64 %[stops], %[stops], %[bit]\n"
        "s_or_b64 %[ins], %[ins], %[bit]\n"
        "s_mov_b32 %[last], %[len]\n"
        "s_cmp_eq_u32 %[len], 0\n"
        "s_cbranch_scc1 2f\n"
        "4:\n"
        "s_or_b64 %[ms], %[ms], %[bit]\n"
        "s_cmp_gt_u32 %[len], 2\n"
        "s_cselect_b64 %[tm], 3, 2\n"
        "s_add_i32 %[l], %[l], %[len]\n"
        "s_add_i32 %[t], %[l], -2\n"
        "s_lshl_b64 %[tm], %[tm], %[t]\n"
        "s_cmp_lt_i32 %[t], 64\n"
        "s_cselect_b64 %[tm], %[tm], 0\n"
        "s_or_b64 %[ins], %[ins], %[tm]\n"
        "s_cmp_lt_i32 %[l], 64\n"
        "s_cbranch_scc1 1b\n"
        "s_branch 9f\n"
        "2:\n"
        "s_add_i32 %[l], %[l], 1\n"
        "s_cmp_lt_i32 %[l], 64\n"
        "s_cbranch_scc1 1b\n"
        "s_branch 9f\n"
        "3:\n"                                     /* a run: no store at the stop itself */
        "s_cmp_lt_i32 %[w], 0\n"
        "s_cbranch_scc1 9f\n"
        "s_and_b32 %[len], %[w], 0x1ff\n"
        "s_lshl_b64 %[bit], 1, %[l]\n"
        "s_or_b64 %[stops], %[stops], %[bit]\n"
        "s_mov_b32 %[last], %[len]\n"
        "s_branch 4b\n"
        "9:\n"
        : [l] "+s"(l), [ins] "+s"(ins), [stops] "+s"(stops), [ms] "+s"(ms), [last] "+s"(last_len),
          [w] "=&s"(w), [len] "=&s"(len), [t] "=&s"(t), [bit] "=&s"(bit), [tm] "=&s"(tm)
        : [W] "v"(W)
        : "scc");
}
