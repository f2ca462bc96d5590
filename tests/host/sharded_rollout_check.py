"""Run under `python -m torch.distributed.run --nproc-per-node W` (tests/test_gpu_sharded_rollout.py): every rank
drives ShardedRollout in the given gather mode for a few passes and checks that EVERY rank-major block of the
receive buffers holds exactly what that rank's shard produces — recomputed locally with a second engine that is
given the peer's global env offset, initial state and actions (results are independent of the sharding).

Backends: `nccl` (RCCL; one rank per GPU, or a single rank) or `gloo` with all ranks sharing cuda:0
(EMEI_BENCH_SHARE_GPU-style rehearsal on a one-GPU box).
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--gather", default="per_chunk")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=16)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--exchange", default="collective")
    ap.add_argument("--slow-consumer", action="store_true",
                    help="a consumer on a third stream reads EVERY receive buffer, slowly (zoo/util.py:54-59 reads every obs): "
                         "under ObsExchange.reading() no buffer may be overwritten before it has been read")
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(a.backend)

    from emei_amd.engine import Engine
    from emei_amd.sharding import ShardedRollout, synthetic_actions, synthetic_init_state

    env, n, T = "CartPoleSwingUp", a.envs, a.horizon
    sr = ShardedRollout(env, n, T, rank=rank, world=world, device=dev, seed=0, gather=a.gather, chunk=a.chunk,
                        force_exchange=True, exchange_algo=a.exchange)
    sr.make_synthetic_inputs()
    # reference engines: one per rank of the job, all on THIS rank's GPU
    refs = []
    for r in range(world):
        e = Engine(env, n, max_episode_steps=1000, device=dev, seed=0, env_index_offset=r * n)
        e.set_state(synthetic_init_state(env, world * n, r * n, (r + 1) * n, 0))
        refs.append((e, synthetic_actions(env, T, n, r, world, e.device, 0)))
    K = sr.chunk
    consumer_stream = torch.cuda.Stream(device=dev)
    seen = []  # (pass, chunk, copy of the receive buffer as the slow consumer saw it)
    spin = torch.ones((1024, 1024), device=f"cuda:{dev}")

    def slow_consumer(p):
        def on_gathered(c, buf):
            with torch.cuda.stream(consumer_stream):
                with sr.xchg.reading(buf):
                    x = spin
                    for _ in range(40):  # ~ms of work in front of the read: the producer is several collectives ahead by then
                        x = (x @ spin) * 1e-3
                    seen.append((p, c, buf.clone(), float(x[0, 0].item() * 0)))
        return on_gathered

    def quick_consumer(p):  # peer_write without --slow-consumer: every chunk's buffer is still copied out under reading()
        def on_gathered(c, buf):
            with torch.cuda.stream(consumer_stream):
                with sr.xchg.reading(buf):
                    seen.append((p, c, buf.clone(), 0.0))
        return on_gathered

    pw = a.exchange == "peer_write"
    copy_all = a.slow_consumer or pw
    for p in range(a.passes):
        last = sr.run_pass(on_gathered=slow_consumer(p) if a.slow_consumer else (quick_consumer(p) if pw else None))
        sr.wait_gathers()
        torch.cuda.synchronize()
        want = [e.rollout(acts, auto_reset=True)[0] for e, acts in refs]  # [T, n, 4] per rank
        torch.cuda.synchronize()
        assert torch.equal(sr.out[0], want[rank]), f"pass {p}: own shard"
        rows = 1 if a.gather == "final" else K
        assert tuple(last.shape) == (world, rows, n, 4), tuple(last.shape)
        # the two receive buffers hold the LAST TWO collectives of the pass
        n_coll = 1 if a.gather == "final" else sr.n_chunks
        # (peer writes: a rank's buffers are written by its PEERS, who may be a launch ahead — the block before the last is only
        # guaranteed while its reader holds it (reading()); the copies taken under reading() are checked below instead)
        for back in range(min(1 if pw else 2, n_coll)):
            buf = sr.xchg.last(back)
            c = n_coll - 1 - back
            for r in range(world):
                blk = want[r][-1:] if a.gather == "final" else want[r][c * K:(c + 1) * K]
                assert torch.equal(buf[r], blk), f"pass {p}: collective {c}, block of rank {r}"
        if copy_all:  # every buffer the consumer copied out holds the block of ITS collective, of every rank
            consumer_stream.synchronize()
            for pp, c, copy, _ in [s for s in seen if s[0] == p]:
                for r in range(world):
                    blk = want[r][-1:] if a.gather == "final" else want[r][c * K:(c + 1) * K]
                    assert torch.equal(copy[r], blk), f"slow consumer, pass {pp}: collective {c}, block of rank {r} was overwritten"
    if copy_all:
        assert len(seen) == a.passes * (1 if a.gather == "final" else sr.n_chunks)
    assert sr.collectives == a.passes * (1 if a.gather == "final" else sr.n_chunks)
    if a.exchange == "peer_write":  # the step-major storage is the same data in global env order
        sm = sr.xchg.step_major[(sr.collectives - 1) & 1]
        assert torch.equal(sm, torch.cat([w[-K:] for w in want], dim=1)), "step-major view"
        assert "peers" in sr.kernel_name
    dist.barrier()
    n_coll, kernel = sr.collectives, sr.kernel_name
    sr.close()  # peer-mapped buffers: unmapped by everybody before their owners free them
    if rank == 0:
        print(f"SHARDED_OK world={world} backend={a.backend} gather={a.gather} chunk={K} collectives={n_coll} kernel={kernel}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
