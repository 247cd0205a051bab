"""The epoch / flag protocol of the peer-write exchange (vlsa_amd/csrc/xchg.hip, vlsa_amd/sharded.py `exchange="ipc"`) as a small
state machine, run under random interleavings of the ranks: what one GPU cannot show (ranks that really progress independently) is
checked here on the CPU -- no deadlock for any schedule, and no buffer is overwritten before its reader is done with it.

Per launch e (slot e & 1) every rank issues, in stream order (pipelined: the tail of launch e - 1 behind the put of launch e):
    PUT_REC(e)   block d: wait ACK_REC[me][slot][d] >= e - 2; write inbox[d][slot][me]; REC[d][slot][me] = e
    WAIT_REC(e)  all d: REC[me][slot][d] >= e;  then the fold reads inbox[me][slot][*]
    PUT_RES(e)   block d: wait ACK_RES[me][slot][d] >= e - 2; write resbox[d][slot][me]; RES[d][slot][me] = e; ACK_REC[d][slot][me] = e
    COLLECT(e)   block o: wait RES[me][slot][o] >= e; read resbox[me][slot][o]; ACK_RES[o][slot][me] = e
A kernel's blocks progress independently; the stream moves on when all of them are done."""
import random

import pytest


class Rank:
    def __init__(self, r, world, launches):
        self.r, self.w = r, world
        ops = []
        for e in range(1, launches + 1):
            ops.append(("PUT_REC", e))
            if e > 1:
                ops += [("WAIT_REC", e - 1), ("PUT_RES", e - 1), ("COLLECT", e - 1)]
        ops += [("WAIT_REC", launches), ("PUT_RES", launches), ("COLLECT", launches)]
        self.ops, self.pc, self.done_blocks = ops, 0, set()


def run_schedule(world, launches, seed):
    rng = random.Random(seed)
    Z = lambda: [[[0] * world for _ in range(2)] for _ in range(world)]  # noqa: E731   [owner of the flag][slot][peer]
    REC, RES, ACK_REC, ACK_RES = Z(), Z(), Z(), Z()
    inbox = [[[0] * world for _ in range(2)] for _ in range(world)]       # epoch of the record lying in inbox[d][slot][src]
    inbox_read = [[[0] * world for _ in range(2)] for _ in range(world)]  # last epoch the owner folded from it
    resbox = [[[0] * world for _ in range(2)] for _ in range(world)]
    resbox_read = [[[0] * world for _ in range(2)] for _ in range(world)]
    ranks = [Rank(r, world, launches) for r in range(world)]
    steps = 0
    while any(k.pc < len(k.ops) for k in ranks):
        progressed = False
        order = list(range(world))
        rng.shuffle(order)
        for r in order:
            k = ranks[r]
            if k.pc >= len(k.ops):
                continue
            op, e = k.ops[k.pc]
            s = e & 1
            blocks = [b for b in range(world) if b not in k.done_blocks]
            rng.shuffle(blocks)
            for b in blocks[:rng.randint(1, len(blocks))]:
                if op == "PUT_REC":
                    if ACK_REC[r][s][b] - (e - 2) < 0:
                        continue
                    assert inbox_read[b][s][r] >= inbox[b][s][r], f"rank {r} overwrites a record owner {b} has not folded (epoch {e})"
                    inbox[b][s][r] = e
                    REC[b][s][r] = e
                elif op == "WAIT_REC":
                    if REC[r][s][b] < e:
                        continue
                    assert inbox[r][s][b] == e, f"owner {r} folds epoch {inbox[r][s][b]} of rank {b}'s record for launch {e}"
                    inbox_read[r][s][b] = e
                elif op == "PUT_RES":
                    if ACK_RES[r][s][b] - (e - 2) < 0:
                        continue
                    assert resbox_read[b][s][r] >= resbox[b][s][r], f"owner {r} overwrites results rank {b} has not collected (epoch {e})"
                    resbox[b][s][r] = e
                    RES[b][s][r] = e
                    ACK_REC[b][s][r] = e
                else:   # COLLECT
                    if RES[r][s][b] < e:
                        continue
                    assert resbox[r][s][b] == e, f"rank {r} collects epoch {resbox[r][s][b]} of owner {b} for launch {e}"
                    resbox_read[r][s][b] = e
                    ACK_RES[b][s][r] = e
                k.done_blocks.add(b)
                progressed = True
            if len(k.done_blocks) == world:
                k.pc += 1
                k.done_blocks = set()
                progressed = True
        steps += 1
        assert progressed, ("deadlock", [(k.r, k.ops[k.pc], sorted(k.done_blocks)) for k in ranks if k.pc < len(k.ops)])
        assert steps < 100_000
    return steps


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_no_deadlock_and_no_overwrite_under_random_interleavings(world):
    for seed in range(120):
        run_schedule(world, launches=7, seed=1000 * world + seed)


def test_a_broken_gate_is_caught_by_the_model(monkeypatch):
    """the model is sharp: WITHOUT the acknowledgement gate of PUT_REC some schedule lets a fast rank overwrite a record its owner
    has not folded yet"""
    caught = 0
    for seed in range(200):
        try:
            run_schedule_early_ack(3, 7, seed)
        except AssertionError as exc:
            if "overwrites a record" in str(exc) or "folds epoch" in str(exc):
                caught += 1
    assert caught > 0


def run_schedule_early_ack(world, launches, seed):
    """run_schedule with ONE change: PUT_REC does not wait for its gate at all"""
    rng = random.Random(seed)
    Z = lambda: [[[0] * world for _ in range(2)] for _ in range(world)]  # noqa: E731
    REC = Z()
    inbox, inbox_read = Z(), Z()
    ranks = [Rank(r, world, launches) for r in range(world)]
    for k in ranks:
        k.ops = [o for o in k.ops if o[0] in ("PUT_REC", "WAIT_REC")]
    for _it in range(20_000):          # (bounded: this broken protocol may also simply stall)
        if not any(k.pc < len(k.ops) for k in ranks):
            return
        progressed = False
        order = list(range(world))
        rng.shuffle(order)
        for r in order:
            k = ranks[r]
            if k.pc >= len(k.ops) or rng.random() < 0.5:
                continue
            op, e = k.ops[k.pc]
            s = e & 1
            for b in range(world):
                if b in k.done_blocks:
                    continue
                if op == "PUT_REC":
                    assert inbox_read[b][s][r] >= inbox[b][s][r], f"rank {r} overwrites a record owner {b} has not folded (epoch {e})"
                    inbox[b][s][r] = e
                    REC[b][s][r] = e
                else:
                    if REC[r][s][b] < e:
                        continue
                    assert inbox[r][s][b] == e, f"owner {r} folds epoch {inbox[r][s][b]} of rank {b}'s record for launch {e}"
                    inbox_read[r][s][b] = e
                k.done_blocks.add(b)
                progressed = True
            if len(k.done_blocks) == world:
                k.pc += 1
                k.done_blocks = set()
