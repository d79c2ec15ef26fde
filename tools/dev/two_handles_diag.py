"""Dev: which buffer of a pass goes wrong first when two model handles run their passes at the same time (round-5 review, item 1).

Two handles (same weights), one host thread each, every round started together from a barrier.  (Written against the round-5 library, where
DFX_PASS_TURN=0 switched the pass gate off; since the handles share the process's streams the enqueue lock cannot be switched off any more — the
tool still shows which buffer differs first should a pass ever come back wrong.)
After a round whose output differs from the solo run, every buffer of the failing handle's workspace is compared with the snapshot of its
solo run: name, differing elements, clips, frames.  The earliest buffer in the dependency order that differs names the kernel.

  python tools/dev/two_handles_diag.py [--B 256] [--T 96000] [--rounds 8] [--streams shared|own] [--env2 K=V ...] [--keep-gate]
"""
import argparse
import os
import sys
import threading

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--T", type=int, default=96000)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--streams", default="shared")
ap.add_argument("--env2", nargs="*", default=[])
ap.add_argument("--keep-gate", action="store_true")
ap.add_argument("--max-dumps", type=int, default=2)
ap.add_argument("--stagger-us", type=int, default=0)
args = ap.parse_args()
if not args.keep_gate:
    os.environ["DFX_PASS_TURN"] = "0"

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from deepfilternet_amd.config import ModelParams  # noqa: E402
from deepfilternet_amd.enhance import enhance, init_df  # noqa: E402
from deepfilternet_amd.state_dict import random_state_dict  # noqa: E402

p = ModelParams.deepfilternet3()
B, T = args.B, args.T
xs = [torch.from_numpy((0.1 * np.random.default_rng(1 + i).standard_normal((B, T))).astype(np.float32)).cuda() for i in range(2)]
models = []
for i in range(2):
    if i == 1:
        for kv in args.env2:
            k, v = kv.split("=", 1)
            os.environ[k] = v
    models.append(init_df(params=p, state_dict=random_state_dict(p, 0), epoch="none")[:2])


def align(n, a):
    return (n + a - 1) // a * a


def layout(ws_ptr):
    """(name, byte offset from ws_ptr, floats, kind) of every buffer: plan_enh + plan_ws of csrc/dfx_model.hip."""
    N, hop, F, E, Fd, C, O = p.fft_size, p.hop_size, 488, p.nb_erb, p.nb_df, p.conv_ch, p.df_order
    Tf = (T + N) // hop
    R = B * Tf
    emb, NO = C * E // 4, 2 * O
    base = align(ws_ptr, 256) - ws_ptr
    out, off = [], 0

    def take_b(name, nbytes, kind):
        nonlocal off
        out.append((name, base + off, nbytes // 4, kind))
        off += align(nbytes, 256)

    take_b("spec", R * F * 8, ("rows", F * 2))
    take_b("spec_e", R * F * 8, ("rows", F * 2))
    take_b("feat_erb", R * E * 4, ("rows", E))
    take_b("feat_spec", R * Fd * 8, ("rows", Fd * 2))
    mbase = base + off
    foff = 0

    def take(name, n, kind):
        nonlocal foff
        if n:
            out.append((name, mbase + foff * 4, n, kind))
        foff += align(n, 64)

    take("e0", R * E * C, ("rows", E * C))
    take("e1", R * (E // 2) * C, ("rows", E // 2 * C))
    take("e2", R * (E // 4) * C, ("rows", E // 4 * C))
    take("e3", R * (E // 4) * C, ("rows", E // 4 * C))
    take("c1", R * (Fd // 2) * C, ("rows", Fd // 2 * C))
    take("emb_in", R * emb, ("rows", emb))
    take("emb", R * emb, ("rows", emb))
    take("xa", R * 256, ("rows", 256))
    take("xb", R * 256, ("rows", 256))
    take("gi", R * 768, ("rows", 768))
    take("xa2", R * 256, ("rows", 256))
    take("xb2", R * 256, ("rows", 256))
    take("gi2", R * 768, ("rows", 768))
    take("demb", R * emb, ("rows", emb))
    take("d3", R * (E // 4) * C, ("rows", E // 4 * C))
    take("d2", R * (E // 2) * C, ("rows", E // 2 * C))
    take("d1", R * E * C, ("rows", E * C))
    take("mask", R * E, ("rows", E))
    take("c0p", R * Fd * NO, ("botf", (O, Tf, Fd * 2)))
    take("xdf", R * 256, ("rows", 256))
    take("coefs", R * Fd * NO, ("botf", (O, Tf, Fd * 2)))
    take("lsnr", R, ("rows", 1))
    take("skp_e", 0, None)
    take("skp_d", R * emb if p.emb_gru_skip == "groupedlinear" else 0, ("rows", emb))
    nl = 1 + (p.emb_num_layers - 1) + p.df_num_layers
    for l in range(8):
        used = l < nl
        take(f"pgi{l}", R * 768 if used else 0, ("rows", 768))
        take(f"py{l}", R * 256 if used else 0, ("rows", 256))
        take(f"ph{l}", B * 256 if used else 0, ("clip", 256))
    return out, Tf


def diff_ws(now, ref, ptr, other=None):
    lay, Tf = layout(ptr)
    for name, boff, n, kind in lay:
        if kind is None or boff + n * 4 > now.numel():
            if kind is not None:
                print(f"    {name}: beyond the workspace (layout mismatch?)")
            continue
        a = now[boff:boff + n * 4].view(torch.int32)
        b = ref[boff:boff + n * 4].view(torch.int32)
        ne = (a != b).nonzero().flatten()
        if ne.numel() == 0:
            continue
        if kind[0] == "rows":
            r = ne // kind[1]
            clip, fr = r // Tf, r % Tf
        elif kind[0] == "botf":
            O_, Tf_, row = kind[1]
            clip = ne // (O_ * Tf_ * row)
            fr = (ne // row) % Tf_
        else:
            clip, fr = ne // kind[1], torch.zeros_like(ne)
        cl = torch.unique(clip).tolist()
        if name == "spec":   # the first buffer of the pass: where exactly, and what stands there
            idx = ne.tolist()
            runs, start, prev = [], idx[0], idx[0]
            for j in idx[1:]:
                if j != prev + 1:
                    runs.append((start, prev))
                    start = j
                prev = j
            runs.append((start, prev))
            for (a0, a1) in runs[:8]:
                r, c0 = divmod(a0, kind[1])
                nv, rv = now[boff:boff + n * 4].view(torch.float32), ref[boff:boff + n * 4].view(torch.float32)
                print(f"      spec run: clip {r // Tf} frame {r % Tf} floats {c0}..{c0 + a1 - a0} (address mod 128 = {(ptr + boff + 4 * a0) % 128}); "
                      f"now {nv[a0:a0 + 4].tolist()} ref {rv[a0:a0 + 4].tolist()}")
                if other is not None:   # the other handle's solo spectrum at the same place / anywhere?
                    ov = other[boff:boff + n * 4].view(torch.float32)
                    same_place = bool(torch.equal(ov[a0:a1 + 1], nv[a0:a1 + 1]))
                    hit = (ov == nv[a0]).nonzero().flatten()[:4].tolist()
                    hit_own = (rv == nv[a0]).nonzero().flatten()[:4].tolist()
                    print(f"        = the other handle's values there: {same_place}; first value found in the other handle's spec at {hit}, in the own solo spec at {hit_own} (this index: {a0})")
        print(f"    {name}: {ne.numel()} of {n} floats differ; clips {cl[:12]}{'...' if len(cl) > 12 else ''} ({len(cl)}), "
              f"groups {sorted(set(c // 16 for c in cl))[:12]}, frames {int(fr.min())}..{int(fr.max())}")


def solo(i):
    y = enhance(models[i][0], models[i][1], xs[i])
    torch.cuda.synchronize()
    models[i][0].check()
    return y.clone(), models[i][0]._ws.clone()


refs = [solo(i) for i in range(2)]
again = [solo(i) for i in range(2)]
for i in range(2):
    print(f"handle {i}: solo output repeatable {torch.equal(refs[i][0], again[i][0])}, workspace repeatable {torch.equal(refs[i][1], again[i][1])}, "
          f"persistent {models[i][0].query(1)} probe {models[i][0].query(2)}")
    if not torch.equal(refs[i][1], again[i][1]):
        print("  solo runs differ in:")
        diff_ws(again[i][1], refs[i][1], models[i][0]._ws.data_ptr())
del again
own = [torch.cuda.Stream() for _ in range(2)] if args.streams == "own" else None
bar = threading.Barrier(2)
outs, errs = [None, None], []


def work(i):
    try:
        bar.wait()
        if i == 1 and args.stagger_us:
            import time
            time.sleep(args.stagger_us * 1e-6)
        if own:
            with torch.cuda.stream(own[i]):
                outs[i] = enhance(models[i][0], models[i][1], xs[i])
        else:
            outs[i] = enhance(models[i][0], models[i][1], xs[i])
    except Exception as e:   # noqa: BLE001
        errs.append((i, repr(e)[:400]))


dumps = 0
bad_rounds = 0
for rnd in range(args.rounds):
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    for i in range(2):
        try:
            models[i][0].check()
        except Exception as e:   # noqa: BLE001
            errs.append((i, "check: " + repr(e)[:400]))
    ok = [outs[i] is not None and torch.equal(outs[i], refs[i][0]) for i in range(2)]
    print(f"round {rnd}: equal {ok} errors {errs}")
    errs.clear()
    if not all(ok):
        bad_rounds += 1
    for i in range(2):
        if not ok[i] and outs[i] is not None and dumps < args.max_dumps:
            dumps += 1
            d = (outs[i] - refs[i][0]).abs()
            bad = (d.amax(dim=1) > 0).nonzero().flatten().tolist()
            cols = (d.amax(dim=0) > 0).nonzero().flatten()
            print(f"  handle {i}: max diff {float(d.max()):.3e}, clips {bad[:16]} ({len(bad)}), samples {int(cols.min())}..{int(cols.max())} ({len(cols)})")
            diff_ws(models[i][0]._ws, refs[i][1], models[i][0]._ws.data_ptr(), refs[1 - i][1])
print(f"SUMMARY streams={args.streams} env2={args.env2} gate={'on' if args.keep_gate else 'off'}: {bad_rounds} of {args.rounds} rounds wrong")
