"""developer probe: LM runs with rejected trials (synth.lever_arm_pair) vs the oracle, iteration by iteration, with the tracked searches verified"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
import torch; torch.cuda.init()
from qn_amd import engine, synth
from oracle import oracle
cases = [(0, 0.1, False, 10, 1e9), (0, 0.03, True, 15, 52.5), (0, 0.1, True, 15, 52.5), (5, 0.1, True, 15, 52.5)]
for seed, rs, scene, K, mcd in cases:
    for force in (0, 12):
        for knobs in ({}, {"verify_track": 1}):
            ctx = engine.Context(8192)
            for k, v in knobs.items(): ctx.debug_set(k, v)
            src, tgt, guess = synth.lever_arm_pair(seed, rot_sigma=rs, scene=scene, n=3000 if scene else 1000)
            mi = 12 if force else 32
            g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(K); g.setMaximumIterations(mi); g.setMaxCorrespondenceDistance(mcd); g.setTransformationEpsilon(0.01); g.setForceIterations(force)
            g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
            g.align(guess.astype(np.float32)); r = g.result_dict()
            o = oracle.GicpOracle(k=K, max_iter=mi, max_corr_dist=mcd, trans_eps=0.01, force_iterations=force)
            o.set_source(src); o.compute_covariances(0); o.set_target(tgt); o.compute_covariances(1)
            ro = o.align(guess.astype(np.float32).astype(np.float64))
            dt, dr = synth.pose_error(r["T"], ro["T"])
            n = min(len(r["trace"]), len(ro["trace"]))
            rel = np.abs(r["trace"][:n, 0] / ro["trace"][:n, 0] - 1)
            bad = np.flatnonzero(rel > 1e-9)
            vm = ctx.debug_get("verify_mismatches") if "verify_track" in knobs else -1
            if not knobs and not force:       # stage by stage at the guess pose
                G = guess.astype(np.float32).astype(np.float64)
                C0, Co0 = g.covariances(0), o.covariances(0); C1, Co1 = g.covariances(1), o.covariances(1)
                H, b, e, corr, sqd = g.linearize(G); Ho, bo, eo, co, so = o.linearize(G)
                dC = np.abs(C0 - Co0).reshape(len(src), -1).max(1)
                print("   stages: max|dC_src| %.2e (points > 1e-9: %d) max|dC_tgt| %.2e corr equal %s sqd equal %s e rel %.2e H rel %.2e" % (
                    dC.max(), int((dC > 1e-9).sum()), np.abs(C1 - Co1).max(), np.array_equal(corr, co), np.array_equal(sqd, so), abs(e / eo - 1), np.abs(H - Ho).max() / np.abs(Ho).max()))
            print("seed %d force %2d knobs %-22s iters %d/%d dt %.2e dr %.2e first y0 mismatch at %s (rel %s) flags_equal %s verify_mismatches %s" % (
                seed, force, json.dumps(knobs), r["iterations"], ro["iterations"], dt, dr, bad[:1], rel[bad[:1]], np.array_equal(r["trace"][:n, 5:], ro["trace"][:n, 5:]), vm))
            ctx.close()
