#!/bin/bash
# round 5: the one-off variant libraries behind profiles/r05_stream_variants.log and r05_stream_clock_ab.log (run on the build host; the
# .so files travel to the GPU box).  Each variant is the PRODUCT's awgn_kernels.hip with a few statements replaced.
set -e
cd "$(dirname "$0")/.."
B="python3 experiments/build_variant.py"
F=awgn_kernels.hip
# the sample kernel's four staging stores, and what replaces them when they are "off" (the values stay alive, nothing is stored)
S1='__builtin_nontemporal_store((u32x4){cnt[0], cnt[1], cnt[2], cnt[3]}, out);'
S2='__builtin_nontemporal_store((u32x4){cnt[4], cnt[5], cnt[6], cnt[7]}, out + 64);'
S3='__builtin_nontemporal_store((u32x4){cnt[0], cnt[1], cnt[2], cnt[3]}, out + 128);'
S4='__builtin_nontemporal_store((u32x4){cnt[4], cnt[5], cnt[6], cnt[7]}, out + 192);'
N1='asm volatile("" :: "v"(cnt[0]), "v"(cnt[1]), "v"(cnt[2]), "v"(cnt[3]), "v"(out));'
N2='asm volatile("" :: "v"(cnt[4]), "v"(cnt[5]), "v"(cnt[6]), "v"(cnt[7]));'
N3='asm volatile("" :: "v"(cnt[0]), "v"(cnt[1]), "v"(cnt[2]), "v"(cnt[3]));'
NOSTORE=("$S1" "$N1" "$S2" "$N2" "$S3" "$N3" "$S4" "$N2")
# the mover: leaves at once / issues no loads / issues no stores (plain mover) / stores non-temporal / DMA non-temporal / a pause behind each store
M='    if (blockIdx.x * ge.per_block >= ge.nunits) return;'; MN='    if (ge.nunits) return;'
D0='    auto dma_unit = [&](const Pos &p, unsigned buf) {
        const unsigned step0 = p.rg * 128;'
D0N='    auto dma_unit = [&](const Pos &p, unsigned buf) {
        if (ge.nunits) return;
        const unsigned step0 = p.rg * 128;'
P1='                    *reinterpret_cast<u32x4 *>(dstw + off) = v;
                    off += goff;'
P1_NONE='                    asm volatile("" :: "v"(v), "v"(dstw + off));
                    off += goff;'
P1_SLEEP='                    *reinterpret_cast<u32x4 *>(dstw + off) = v;
                    __builtin_amdgcn_s_sleep(2);
                    off += goff;'
P1_NT_SLEEP='                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dstw + off));
                    __builtin_amdgcn_s_sleep(2);
                    off += goff;'
MS='*reinterpret_cast<u32x4 *>(dstw + off) = v;'; MSN='__builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dstw + off));'
DA='(lds_void_ptr)(uintptr_t)(rawb + k * 256), 16, 0, 0);'; DAN='(lds_void_ptr)(uintptr_t)(rawb + k * 256), 16, 0, 2);'
SB='        const char *const sb = reinterpret_cast<const char *>(stage) + ((wabs * L + step0) * 128 + p.q8 * 8) * 16;'
# per-wave stamps OUTSIDE the sample kernel's loop (cycles per wave and the clock they ran at: experiments/clock_ab.py)
K1='    const unsigned long long LG = wave * 64 + lane;
    __builtin_amdgcn_s_setprio(3);
    uint32_t a[256], b[256], pa[256], pb[256], cnt[8];
    if constexpr (SMALL) {'
K1N='    const unsigned long long LG = wave * 64 + lane;
    __builtin_amdgcn_s_setprio(3);
    const unsigned long long var_t0 = __builtin_amdgcn_s_memtime(), var_r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t a[256], b[256], pa[256], pb[256], cnt[8];
    if constexpr (SMALL) {'
K2='        out += 256;
    }
}

int awgn256_planes_launch('
K2N='        out += 256;
    }
    if (lane == 0 && wave < 1024) {
        bbb_var_stamps[4 * wave] = var_t0; bbb_var_stamps[4 * wave + 1] = __builtin_amdgcn_s_memtime();
        bbb_var_stamps[4 * wave + 2] = var_r0; bbb_var_stamps[4 * wave + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace bbb
extern "C" int bbb_var_read_stamps(unsigned long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bbb::bbb_var_stamps), sizeof(unsigned long long) * 4 * 1024); }
namespace bbb {

int awgn256_planes_launch('
K3='template <bool SMALL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
awgn256_planes_kernel('
K3N='__device__ unsigned long long bbb_var_stamps[4 * 1024];
template <bool SMALL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
awgn256_planes_kernel('
STAMPS=("$K1" "$K1N" "$K2" "$K2N" "$K3" "$K3N")
$B nostore $F "${NOSTORE[@]}"
$B plainst $F "$S1" '*out = (u32x4){cnt[0], cnt[1], cnt[2], cnt[3]};' "$S2" 'out[64] = (u32x4){cnt[4], cnt[5], cnt[6], cnt[7]};' "$S3" 'out[128] = (u32x4){cnt[0], cnt[1], cnt[2], cnt[3]};' "$S4" 'out[192] = (u32x4){cnt[4], cnt[5], cnt[6], cnt[7]};'
$B shallow $F 'wait_vm((more1 ? NDMA : 0u) + (fast1 ? NST : 0u));' 'wait_vm(more1 ? NDMA : 0u);'
$B mvnt $F "$MS" "$MSN"
$B paced $F "$P1" "$P1_SLEEP"
$B nomover $F "$M" "$MN"
$B nomover_nostore $F "$M" "$MN" "${NOSTORE[@]}"
$B nodma $F "$D0" "$D0N"
$B nomvstore $F "$P1" "$P1_NONE"
$B mvnt_dmant $F "$MS" "$MSN" "$DA" "$DAN"
$B mvnt_paced $F "$P1" "$P1_NT_SLEEP"
$B mvnt_dmant_paced $F "$P1" "$P1_NT_SLEEP" "$DA" "$DAN"
$B st $F "${STAMPS[@]}"
$B st_nostore $F "${STAMPS[@]}" "${NOSTORE[@]}"
$B st_nomover $F "${STAMPS[@]}" "$M" "$MN"
for mb in 64 16; do
  $B mallread$mb $F "$SB" "        const char *const sb = reinterpret_cast<const char *>(stage) + ((((wabs * L + step0) * 128 + p.q8 * 8) * 16) & ((${mb}ull << 20) - 1));"
done
# (split8: every 16-byte staging store as two 8-byte stores -- twice the store instructions, the same bytes; built by hand, see profiles/r05_stream_variants.log)
# the mover's ACCESS PATTERN: every unit written / read as one contiguous 32 KiB block (wrong places, the same bytes)
P1C='                    *reinterpret_cast<u32x4 *>(dst + ((unsigned long long)(u0 + it) * 32768ull + ((k * 4 + wv) * 8 + lq) * 128 + c2 * 16) % (nbytes_ & ~0xffffull)) = v;
                    off += goff;'
R1='                __builtin_amdgcn_global_load_lds((const void *)(pl + ((k & 3) * 32 * 2048 + (k >> 2) * 1024)),'
R1C='                __builtin_amdgcn_global_load_lds((const void *)(reinterpret_cast<const char *>(stage) + (((wabs * ge.ngroups * 8 + p.q8 * ge.ngroups + p.rg) * 32768ull) % (1ull << 30)) + wv * 8192 + k * 1024 + lane * 16),'
$B mvcontig $F "$P1" "$P1C"
$B rdcontig $F "$R1" "$R1C"
$B bothcontig $F "$P1" "$P1C" "$R1" "$R1C"
