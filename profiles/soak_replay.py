"""Soak of the evaluation replay (csrc/replay.cpp): twin graphs built from one seed, one evaluated with replay on, the other
with replay off, driven through the same random sequence of re-evaluations, re-plugged cables (the same edge connected again:
the case that replays) and real edits (Mix type, rewiring, removed edges, use_cache, in-place materialisation).  After every
step: the same result bit for bit (or the same error), the same node states, the same nodes holding slot data, the same
changed set.     python profiles/soak_replay.py [graphs] [first seed]"""
import faulthandler, json, os, sys, time
faulthandler.enable()
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanter_core_amd as kc
from oracle import oracle as orc
from util import bit_equal
import test_gpu_fuzz_graphs as fz

kc.init(0)
n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 500
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
OPS = fz.OPS
bad = 0
t0 = time.time()
r0 = kc.stats_counter("replayed_evaluations")


def evaluate(lg, target, replay):
    kc.set_option("replay", replay)
    try:
        sds = lg.await_clean(target).node_slot_datas(target)
        out = [(s.slot_id, s.image.is_rgba(), s.image.planes()) for s in sorted(sds, key=lambda s: s.slot_id)]
    except kc.TexProError as e:
        out = "error %s" % e.kind
    ids = sorted(lg.node_ids())
    snap = ([(int(i), lg.node_state(i)) for i in ids], [(int(i), len(lg.node_slot_datas(i))) for i in ids], sorted(int(x) for x in lg.changed_consume()))
    return out, snap


def same(a, b):
    if isinstance(a, str) or isinstance(b, str):
        return a == b
    if len(a) != len(b):
        return False
    for (s1, r1, p1), (s2, r2, p2) in zip(a, b):
        if s1 != s2 or r1 != r2 or len(p1) != len(p2) or not all(bit_equal(x, y) for x, y in zip(p1, p2)):
            return False
    return True


def build_simple(seed, info):
    """A replay-friendly graph: sources of ONE size, a mostly linear run of Mix nodes (every one continues the previous result
    and takes a source, a constant or an invert against it): usually one fused launch, which is what gets recorded."""
    rng = np.random.default_rng(seed)
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 70))
    outs = []
    for eid in range(int(rng.integers(1, 4))):
        planes = [(rng.random((h, w), dtype=np.float32) * np.float32(1.6) - np.float32(0.3)).astype(np.float32) for _ in range(4)]
        lg.embed_slot_data_with_id(kc.SlotData(0, 0, kc.SlotImage.from_planes(planes)), eid)
        outs.append((lg.add_node(kc.Node.new(kc.NodeType.Embed(eid))), 0, "R"))
    vals = []
    for _ in range(int(rng.integers(1, 3))):
        v = lg.add_node(kc.Node.new(kc.NodeType.Value(float(np.float32(rng.random())))))
        c = lg.add_node(kc.Node.new(kc.NodeType.CombineRgba))
        for s_ in range(3):
            lg.connect(v, c, 0, s_)
        vals.append((c, 0, "R"))
    prev = outs[0]
    nodes = []
    for _ in range(int(rng.integers(2, 24))):
        n = lg.add_node(kc.Node.new(kc.NodeType.Mix(kc.MixType.parse(OPS[rng.integers(len(OPS))]))))
        other = (outs + vals)[rng.integers(len(outs) + len(vals))]
        first_out = len(outs)
        if rng.random() < 0.5:
            lg.connect(prev[0], n, prev[1], 0)
            lg.connect(other[0], n, other[1], 1)
        else:
            lg.connect(other[0], n, other[1], 0)
            lg.connect(prev[0], n, prev[1], 1)
        nodes.append((n, "Mix", "XX", first_out))
        prev = (n, 0, "X")
        outs.append(prev)
    info["nodes"], info["outs"] = nodes, outs + vals
    return lg, None, [prev[0]]


for g in range(n_graphs):
    seed = 0x5EA70000 + seed0 + g
    rng = np.random.default_rng(seed)
    info1, info2 = {}, {}
    builder = build_simple if g % 4 else (lambda sd, inf: fz._build(kc, orc, sd, inf))
    lg1, _, req = builder(seed, info1)
    lg2, _, _ = builder(seed, info2)
    if not req:
        continue
    # half of the graphs: a pointwise-only request is likelier to qualify for recording when resize policies are the default ones
    target = req[0]
    nodes, outs = info1["nodes"], info1["outs"]
    last_edge = None
    for step in range(12):
        r = rng.random()
        if r < 0.55:
            edges = lg1.edges()
            if edges:
                # usually the cable that was re-plugged last time: three identical steps in a row are what replays
                e = last_edge if (last_edge is not None and rng.random() < 0.8) else edges[rng.integers(len(edges))]
                last_edge = e
                for lg in (lg1, lg2):
                    lg.connect(e.output_id, e.input_id, e.output_slot, e.input_slot)
        elif r < 0.7:
            pass  # plain re-evaluation (everything Clean: nothing to do either way)
        else:
            n, kind, slots, first_out = nodes[rng.integers(len(nodes))]
            edit = rng.integers(5)
            if edit == 0 and kind == "Mix":
                mt = kc.MixType.parse(OPS[rng.integers(len(OPS))])
                for lg in (lg1, lg2):
                    lg.set_mix_type(n, mt)
            elif edit == 1:
                slot = int(rng.integers(len(slots)))
                cands = [o for o in outs[:first_out] if slots[slot] == "X" or o[2] in (slots[slot], "X")]
                if cands:
                    src = cands[rng.integers(len(cands))]
                    for lg in (lg1, lg2):
                        lg.connect(src[0], n, src[1], slot)
            elif edit == 2:
                slot = int(rng.integers(len(slots)))
                for lg in (lg1, lg2):
                    try:
                        lg.disconnect_slot(n, kc.Side.Input, slot)
                    except kc.TexProError:
                        pass
            elif edit == 3:
                for lg in (lg1, lg2):
                    lg.use_cache = not lg.use_cache
            else:
                for lg in (lg1, lg2):
                    try:
                        for sd in lg.node_slot_datas(n):
                            sd.image.materialize()
                    except kc.TexProError:
                        pass
        o1, s1 = evaluate(lg1, target, 1)
        o2, s2 = evaluate(lg2, target, 0)
        if not same(o1, o2) or s1 != s2:
            bad += 1
            print("MISMATCH graph seed %d step %d: results equal %s, snapshots equal %s" % (seed, step, same(o1, o2), s1 == s2), flush=True)
            break
    if g % 100 == 99:
        print("%d graphs, %d replayed evaluations, %d bad, %.0f s" % (g + 1, kc.stats_counter("replayed_evaluations") - r0, bad, time.time() - t0), flush=True)
kc.set_option("replay", 1)
print("soak_replay: %d graphs x 12 steps, %d replayed evaluations, %d mismatches, %.0f s" % (n_graphs, kc.stats_counter("replayed_evaluations") - r0, bad, time.time() - t0))
sys.exit(1 if bad else 0)
