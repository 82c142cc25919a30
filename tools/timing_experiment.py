"""TIMING-ONLY experiments on the step's launch program (results are INVALID by construction; round 6: moved out of Engine, where
a stray environment variable could silence ops of a real training job - VERDICT r5 item 6 / ADVICE r5).

    python tools/timing_experiment.py --skip "<substr>[,<substr>...]" [-- <bench.py arguments>]
    python tools/timing_experiment.py --defer "<substr>[,<substr>...]" [-- <bench.py arguments>]

--skip   ops whose name contains one of the substrings launch NOTHING (their events / waits stay): the marginal value of an op group
         in the step (profiles/r05/skip_ops_marginal_value.txt)
--defer  the weight-gradient-lane ops whose names contain one of the substrings run at the FRONT of the step, under the forward, as a
         deferred-weight-gradient schedule would run them (profiles/r05/defer_experiment.txt)

The patches live in THIS process only: the tool wraps Engine._op / Engine.step_pieces, stamps the engine so that bench.py prints
config.options.TIMING_EXPERIMENT_INVALID in its JSON line, and then runs bench.main() with the remaining arguments. bench.py itself refuses to
run when the old variables MPMAE_SKIP_OPS / MPMAE_DEFER_EXPERIMENT are set."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def install(skip=(), defer=()):
    import mmearth_train_amd  # noqa: F401
    from mmearth_train_amd import engine as E
    skip, defer = [t for t in skip if t], [t for t in defer if t]
    E.TIMING_EXPERIMENT = dict(skip=skip, defer=defer)
    op0, pieces0 = E.Engine._op, E.Engine.step_pieces

    def _op(self, lst, name, fn, *args, **kw):
        if skip and any(t in name for t in skip):
            kw.setdefault("kind", fn.__name__)
            fn, args = (lambda *a: 0), ()
        return op0(self, lst, name, fn, *args, **kw)

    def step_pieces(self, bwd_segments=None, **kw):
        if not defer or bwd_segments is not None:
            return pieces0(self, bwd_segments, **kw)
        moved = [op for op in self.bwd_ops if op[3]["lane"] == 1 and any(t in op[0] for t in defer)]
        dead = {m[3]["signal"] for m in moved if m[3]["signal"]}
        gone = {id(op) for op in moved}
        keep = [(op[0], op[1], op[2], dict(op[3], wait=tuple(w for w in op[3]["wait"] if w not in dead))) for op in self.bwd_ops if id(op) not in gone]
        saved, self.bwd_ops = self.bwd_ops, keep
        try:
            pieces = pieces0(self, None, **kw)
        finally:
            self.bwd_ops = saved
        fwd = pieces[0]
        front = max(i for i, op in enumerate(fwd[:14]) if op[3]["lane"] == 1)      # behind the side lane's own front (zero fill, weight staging, poolings)
        pieces[0] = fwd[:front + 1] + [(n, f, a, dict(m, wait=(), signal=None)) for n, f, a, m in moved] + fwd[front + 1:]
        print(f"[defer experiment] {len(moved)} ops moved under the forward: {[m[0] for m in moved]}", file=sys.stderr)
        return pieces

    E.Engine._op, E.Engine.step_pieces = _op, step_pieces


def main():
    argv = sys.argv[1:]
    rest = []
    if "--" in argv:
        i = argv.index("--")
        argv, rest = argv[:i], argv[i + 1:]
    skip, defer = [], []
    while argv:
        k = argv.pop(0)
        if k == "--skip":
            skip = argv.pop(0).split(",")
        elif k == "--defer":
            defer = argv.pop(0).split(",")
        else:
            raise SystemExit(f"unknown argument {k} (bench.py arguments go behind --)")
    install(skip, defer)
    import bench
    sys.argv = ["bench.py"] + rest
    bench.main()


if __name__ == "__main__":
    main()
