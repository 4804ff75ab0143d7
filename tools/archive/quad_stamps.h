// s_memtime phase stamps for estep_quad.h (development builds only: tools/phase_stamps_quad.py compiles a COPY of the
// package with -DPYLDA_QUAD_STAMPS=1; the library never includes this file otherwise).  Every stamp drains the wavefront's
// LDS queue, so absolute times are 10-15 % high; the split between the phases is what counts.  The per-phase sums of
// every wavefront, its HW_ID and LDS_ALLOC registers are written over the document's gamma row (needs K >= 16 x wavefronts).
#pragma once
#define QUAD_STAMPS_BEGIN()                                                                     \
    long long stamp_acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                       \
    long long stamp_prev = __builtin_amdgcn_s_memtime();                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define QUAD_STAMP(j)                                                          \
    do {                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
        const long long now_ = __builtin_amdgcn_s_memtime();                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
        stamp_acc[j] += now_ - stamp_prev;                                     \
        stamp_prev = now_;                                                     \
    } while (0)
#define QUAD_STAMPS_DUMP()                                                                      \
    do {                                                                                        \
        QUAD_STAMP(10);                                                                         \
        if (lane == 0) {                                                                        \
            unsigned hw_id, lds_alloc;                                                          \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));                 \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds_alloc));         \
            double* dbg = p.gamma + (size_t)doc * K + wave * 16;                                \
            for (int j = 0; j < 11; ++j) dbg[j] = (double)stamp_acc[j];                         \
            dbg[11] = (double)it;                                                               \
            dbg[12] = (double)hw_id;                                                            \
            dbg[13] = (double)lds_alloc;                                                        \
            dbg[14] = (double)stamp_acc[11];                                                    \
            dbg[15] = (double)stamp_acc[12];                                                    \
        }                                                                                       \
    } while (0)
