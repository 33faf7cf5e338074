/*
 * TEST INFRASTRUCTURE ONLY: the GLZ walk's loop over the ordinary stops of a quarter in plain
 * C, statement for statement what spice_amd/csrc/target/spice_target_glz_walk.h does in GCN
 * assembly.  The simulator build runs this one; what it proves is the algorithm around the
 * loop, not the assembly, which only the GPU tests run.
 */
__device__ __forceinline__ void glz_fast_stops(uint32_t W, int &l, uint64_t &ins, uint64_t &stops,
                                               uint64_t &ms, int &last_len)
{
    while (l < 64) {
        const uint32_t w = glz_readlane(W, l);
        if (w & GLZ_W_SLOW) {
            break;
        }
        const uint64_t bit = 1ull << l;
        int len;
        if (w & GLZ_W_RUN) {
            len = (int)(w & 0x1ffu); /* a run stores nothing at its first position */
        } else {
            const int d1 = (int)(w & 63u), d2 = (int)((w >> GLZ_W_D2_SHIFT) & 63u);
            len = ((ins >> d2) & 1) ? (int)((w >> GLZ_W_L2_SHIFT) & 31u) : (int)((w >> GLZ_W_LF_SHIFT) & 31u);
            len = ((ins >> d1) & 1) ? (int)((w >> GLZ_W_L1_SHIFT) & 31u) : len;
            if (len >= GLZ_W_LEN_MAXED) {
                break;
            }
            ins |= bit;
        }
        stops |= bit;
        if (len == 0) {
            l++;
        } else {
            const uint64_t tm = len > 2 ? 3ull : 2ull;
            const int sh0 = l + len - 2;
            ms |= bit;
            ins |= sh0 < 64 ? tm << sh0 : 0ull;
            l += len;
        }
        last_len = len;
    }
}
