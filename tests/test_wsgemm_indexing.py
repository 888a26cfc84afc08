"""Host-side walk of wsgemm_kernel's index arithmetic (stablediffusion_amd/csrc/wsgemm.hip): every residual row a block
prefetches lies inside its own run of tiles, so no load of the residual tensor can pass row M -- the question the
round-2 GPU fault (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in the RES variant, gpurun_out/ws1.log) left open
(VERDICT r2 #9).  Mirrors the kernel: XCD x owns M tiles [x * tiles_m / 8, (x + 1) * tiles_m / 8), its 32 blocks split
the (n tile, m tile) list into contiguous runs; a run of T tiles is FIRST, STEADY x (T - 1), LAST tile positions; in
position t step KT prefetches chunk (KT + R) % NK of tile t - 1 + (KT + R) // NK, except where the mode has no such
tile (FIRST: no tile before; LAST: no tile after)."""
import pytest

BM, NK, R, GRID = 128, 5, 2, 256


def residual_rows(M, Cout, BN=160):
    tiles_m, tiles_n = M // BM, Cout // BN
    nch = BN // 2 // 16
    nblk = GRID // 8
    for blk in range(GRID):
        xcd, kblk = blk & 7, blk >> 3
        mt_lo, mt_hi = xcd * tiles_m // 8, (xcd + 1) * tiles_m // 8
        nm = mt_hi - mt_lo
        L = tiles_n * nm
        lo, hi = kblk * L // nblk, (kblk + 1) * L // nblk
        while lo < hi:
            tn, mo = divmod(lo, nm)
            T = min(nm - mo, hi - lo)
            mt0 = mt_lo + mo
            lo += T
            for t in range(T + 1):                       # tile positions 0 .. T (T = the LAST epilogue)
                mode = 0 if t == 0 else (2 if t == T else 1)
                for kt in range(NK):
                    if (kt + R) % NK >= nch:
                        continue
                    nxt = (kt + R) // NK == 1
                    if not ((mode != 2) if nxt else (mode != 0)):
                        continue
                    te = t - 1 + (kt + R) // NK
                    yield blk, T, te, (mt0 + te) * BM, (mt0 + te + 1) * BM - 1


@pytest.mark.parametrize("M,Cout", [(32768, 320), (12288, 320), (8192, 640), (15360, 960), (1024, 320)])
def test_every_residual_prefetch_stays_inside_its_run(M, Cout):
    n = 0
    for blk, T, te, row_lo, row_hi in residual_rows(M, Cout):
        assert 0 <= te < T, (blk, T, te)
        assert 0 <= row_lo and row_hi < M, (blk, row_lo, row_hi, M)
        n += 1
    assert n > 0
