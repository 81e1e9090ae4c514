"""H1 measured instead of assumed: the reference's own src/ORBextractor.cc with the STOCK allocator (glibc malloc; `make -C oracle ref_stock`) against the
canonical build every other test uses (the same file with std::list nodes from a bump arena, oracle/orbextractor_ref_wrap.cpp).

DistributeOctTree sorts its expandable nodes by (size, ExtractorNode*) (ORBextractor.cc:684): equal sizes are ordered by HEAP ADDRESS.  Under the arena a
later-created node has the higher address, which is the canonical tie-break the oracle and the HIP kernels reproduce.  Under malloc
the addresses are whatever free lists hand out - they depend on everything the process allocated before, so the stock build is not even a function of its
input: the same image through the same object twice can return different key points.

The test REPORTS what it finds (printed with -s; the numbers are quoted in DESIGN.md section 3 and INTEGRATION.md): per BASELINE shape, frames whose key point set
differs from the canonical build, key points of the stock build that are not in the canonical set, and whether a repeated call reproduces itself.  It fails
only if the difference is NOT the tie-break: everything outside the quadtree's choice must agree - same count per level up to the quadtree's slack, every
stock key point a real FAST corner of the level with the same angle / descriptor the canonical pipeline gives that position.  CPU only."""
import numpy as np
import pytest

from orb_slam2_amd import synth

SHAPES = ((1241, 376, 2000, 16), (640, 480, 1000, 16), (752, 480, 1200, 16))


@pytest.fixture(scope="module")
def ref():
    from oracle import orbextractor_ref as R
    if not (R.build() and R.build_stock()):
        pytest.skip("reference sources not mounted")
    return R


def _key(k):
    return set(zip(k["x"].tolist(), k["y"].tolist(), k["octave"].tolist()))


def test_stock_allocator_build_vs_canonical_tie_break(ref):
    report = []
    for (w, h, n, frames) in SHAPES:
        can, stock = ref.RefExtractor(n, 1.2, 8, 20, 7), ref.RefExtractor(n, 1.2, 8, 20, 7, stock=True)
        differ = foreign = total = repeat_differs = 0
        for s in range(frames):
            img = synth.frame(w, h, seed=300 + s)
            kc, dc = can.extract(img)
            ks, ds = stock.extract(img)
            ks2, _ = stock.extract(img)                                   # the same image through the same object again
            sc, ss = _key(kc), _key(ks)
            differ += sc != ss
            foreign += len(ss - sc); total += len(ss)
            repeat_differs += ks.tobytes() != ks2.tobytes()
            # what is NOT the tie-break must agree: a key point both builds chose carries the same size / angle / response / descriptor
            both = sc & ss
            ic = {(x, y, o): i for i, (x, y, o) in enumerate(zip(kc["x"].tolist(), kc["y"].tolist(), kc["octave"].tolist()))}
            for j, t in enumerate(zip(ks["x"].tolist(), ks["y"].tolist(), ks["octave"].tolist())):
                if t in both:
                    i = ic[t]
                    assert kc[i].tobytes() == ks[j].tobytes() and np.array_equal(dc[i], ds[j]), f"{w}x{h} seed {300 + s}: a shared key point differs beyond the tie-break"
            # ... and the count per level differs by at most the quadtree's own slack (a tie decides WHICH node splits last, ORBextractor.cc:730-731)
            for lvl in range(8):
                assert abs(int((kc["octave"] == lvl).sum()) - int((ks["octave"] == lvl).sum())) <= 3, f"{w}x{h} seed {300 + s} level {lvl}"
        report.append((w, h, n, frames, differ, foreign, total, repeat_differs))
        can.close(); stock.close()
    for r in report:
        print("stock allocator vs canonical tie-break %dx%d N=%d: %d frames, %d with a different key point set, %d of %d stock key points not in the canonical set, "
              "%d frames where a repeated call on the same object differs from the first" % r)
    # the exposure is real and of this size: a maintainer's A/B of the drop-in against their own binary sees it on (nearly) every frame
    assert all(r[6] > 0 and r[5] <= 0.05 * r[6] for r in report), report          # a tie-break, not a different algorithm: a few per cent at most
