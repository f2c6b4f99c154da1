import sys
def edit(s):
    n=[0]
    def rep(old,new,count=1):
        nonlocal s
        assert s.count(old)==count, (s.count(old), old[:80])
        s=s.replace(old,new); n[0]+=1
    rep('''// LDS (dynamic, ONE array: 2 x 32 KiB raw + 32 KiB tile; the shaping mover adds its 32 KiB table and 4 KiB of window words):''',
        '''// LDS (dynamic, ONE array: 2 x 32 KiB raw; the TILE takes the place of the raw buffer it was made from -- phase 1 reads its raw
// data into registers, a barrier, then writes the tile over it: 64 KiB instead of 96 (round 5), so that TWO blocks fit a CU; the
// shaping mover adds an 8 KiB table and 4 KiB of window words: 76 KiB instead of 132):''')
    rep('''constexpr unsigned kUnplaneRaw = 32 * 1024, kUnplaneLds = 3 * 32 * 1024, kUnplaneLdsTx = kUnplaneLds + 32 * 1024 + 2 * 2048;''',
        '''constexpr unsigned kUnplaneRaw = 32 * 1024, kUnplaneLds = 2 * 32 * 1024, kUnplaneLdsTx = kUnplaneLds + 8 * 1024 + 2 * 2048;''')
    rep('''    uint32_t *const tile = lds + 2 * (kUnplaneRaw / 4);
''','')
    rep('''    // phases of one 8-bit window) is widened to T10[q10][e]: the 16 shaped samples of a piece for each value of its 10-bit
''','''    // phases of one 8-bit window) is widened to T9[q9][e]: EIGHT shaped samples for each value of a 9-bit data window -- samples
    // 0..7 of a piece see the window shifted by 0 or 1 data bits, samples 8..15 the same pattern one bit further, so one 8 KiB
    // table serves both halves (round 4 held 16 samples per 10-bit window: 32 KiB, the difference between one and two blocks per CU)
''')
    rep('''    uint32_t *const T10 = lds + kUnplaneLds / 4;              // [1024 windows][8 words = 16 samples]
    uint32_t *const winb = lds + kUnplaneLds / 4 + 8192;      // [2][256 generators][2 words]: the units' data-bit windows, by DMA''',
        '''    uint32_t *const T9 = lds + kUnplaneLds / 4;               // [512 windows][4 words = 8 samples]
    uint32_t *const winb = lds + kUnplaneLds / 4 + 2048;      // [2][256 generators][2 words]: the units' data-bit windows, by DMA''')
    rep('''        uint16_t *const TT = reinterpret_cast<uint16_t *>(tile);      // (the tile is idle until the first unit's phase 1)''',
        '''        uint16_t *const TT = reinterpret_cast<uint16_t *>(lds);       // (raw buffer 0 is idle until the first unit's DMA, issued below)''')
    rep('''        for (unsigned p = tid; p < 1024 * 8; p += 256) {
            const unsigned q10 = p >> 3, e = 2 * (p & 7);
            const uint32_t lo = TT[(((q10 >> ((tx.c0 + e) >> 3)) & 0xffu) << 3) + (e & 7)];
            const uint32_t hi = TT[(((q10 >> ((tx.c0 + e + 1) >> 3)) & 0xffu) << 3) + ((e + 1) & 7)];
            T10[p] = lo | (hi << 16);
        }''','''        for (unsigned p = tid; p < 512 * 4; p += 256) {
            const unsigned q9 = p >> 2, e = 2 * (p & 3);
            const uint32_t lo = TT[(((q9 >> ((tx.c0 + e) >> 3)) & 0xffu) << 3) + (e & 7)];
            const uint32_t hi = TT[(((q9 >> ((tx.c0 + e + 1) >> 3)) & 0xffu) << 3) + ((e + 1) & 7)];
            T9[p] = lo | (hi << 16);
        }''')
    rep('''            const uint32_t *raw = lds + buf * (kUnplaneRaw / 4) + (qs * 8 + l8) * 4;
            uint32_t Z[4][8];''','''            const uint32_t *raw = lds + buf * (kUnplaneRaw / 4) + (qs * 8 + l8) * 4;
            uint32_t *const tile = lds + buf * (kUnplaneRaw / 4);      // the tile takes the raw buffer's place
            uint32_t Z[4][8];''')
    rep('''#pragma unroll
            for (unsigned s = 0; s < 4; s++) planes8_to_bytes(Z[s]);
#pragma unroll
            for (unsigned i = 0; i < 8; i++) {
                uint32_t z[4] = {Z[0][i], Z[1][i], Z[2][i], Z[3][i]};''','''            // every thread holds its raw data in registers before anybody writes the tile over them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (unsigned s = 0; s < 4; s++) planes8_to_bytes(Z[s]);
#pragma unroll
            for (unsigned i = 0; i < 8; i++) {
                uint32_t z[4] = {Z[0][i], Z[1][i], Z[2][i], Z[3][i]};''')
    rep('''        // ---- phase 2: row (4 k + wv) * 8 + lq = generator''','''        const uint32_t *const tile = lds + buf * (kUnplaneRaw / 4);
        // ---- phase 2: row (4 k + wv) * 8 + lq = generator''')
    rep('''            const uint32_t idx_mask = tx.use_bits ? 0x7fe0u : 0u;            // (no data bits: every window reads as 0)''',
        '''            const uint32_t idx_mask = tx.use_bits ? 0x1ff0u : 0u;            // (no data bits: every window reads as 0)''')
    rep('''                // the piece's 10-bit data window (bit j = data bit M0 - 7 + j) selects its row of shaped samples
                const uint32_t wk = (uint32_t)((((unsigned long long)ww[1] << 32) | ww[0]) >> sh);
                const char *const row = reinterpret_cast<const char *>(T10) + ((wk << 5) & idx_mask);
                const u32x4 A = *reinterpret_cast<const u32x4 *>(row);
                const u32x4 B = *reinterpret_cast<const u32x4 *>(row + 16);''','''                // the piece's 10-bit data window (bit j = data bit M0 - 7 + j): its low nine bits select the row of samples 0..7, its
                // high nine that of samples 8..15
                const uint32_t wk = (uint32_t)((((unsigned long long)ww[1] << 32) | ww[0]) >> sh);
                const u32x4 A = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(T9) + ((wk << 4) & idx_mask));
                const u32x4 B = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(T9) + ((wk << 3) & idx_mask));''')
    rep('''    uint64_t blocks = (uint64_t)ncu * (uint64_t)env_knob("BBB_UNPLANE_BLOCKS_PER_CU", 1);''','''    // two blocks per CU (round 5: the LDS of a block went from 96 / 132 KiB to 64 / 76).  Beside the big form of the sample kernel a
    // SIMD has registers for one guest wave, and the second block waits its turn; beside the small form (the transmitter's) and when
    // the movers run alone (the drain of a short run) both are resident
    uint64_t blocks = (uint64_t)ncu * (uint64_t)env_knob("BBB_UNPLANE_BLOCKS_PER_CU", 2);''')
    return s, n[0]
for f in sys.argv[1:]:
    s=open(f).read()
    s,n=edit(s)
    open(f,'w').write(s)
    print(f, n, "edits")
