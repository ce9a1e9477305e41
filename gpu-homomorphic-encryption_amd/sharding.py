"""Multi-GPU plumbing for the batch-sharded path (one process per GPU, no data-path collective).

The polymul path partitions into independent units (polynomial x limb), so ranks never exchange
payload data: torch.distributed (RCCL on the GPU box, gloo in the CPU rehearsal) is used only for the
barriers that bracket a timed region and for the max-over-ranks of the timings / a checksum gather."""
import os


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(total, rank, world):
    """Contiguous block of `total` units owned by `rank`: sizes differ by at most one, blocks are
    ordered by rank and cover [0, total) exactly once (strong-scaling split of a fixed batch)."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def limb_shard(num_limbs, rank, world):
    """Limb split (secondary partitioning; reference docs/ARCHITECTURE.md:499-512 "Distribute RNS components across GPUs"):
    the limbs l = rank (mod world) of EVERY polynomial.  Empty when world > num_limbs leaves this rank nothing.  Valid for the work that
    never mixes limbs (transforms, pointwise products, the tensor product); relinearisation / key switching / base conversion read every
    limb to produce each limb and stay on the batch split."""
    if world < 1 or not (0 <= rank < world) or num_limbs < 0:
        raise ValueError("bad shard arguments")
    return list(range(rank, num_limbs, world))


def choose_split(batch, num_limbs, world, cross_limb=False):
    """Which partitioning a job of `batch` polynomials x `num_limbs` limbs takes on `world` GPUs: the batch split whenever
    every rank gets a polynomial; the limb split for small batches of limb-independent work; otherwise replicas do not help."""
    if batch >= world:
        return "batch"
    if not cross_limb and num_limbs >= world:
        return "limb"
    return "batch" if batch > 0 else "none"


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_process_group(backend=None):
    """Rendezvous from the torchrun environment (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE)."""
    import torch.distributed as dist
    rank, world, local_rank = env_rank_world()
    if world == 1 and os.environ.get("FHE_BENCH_FORCE_DIST") != "1":
        return None
    if world == 1:                              # FHE_BENCH_FORCE_DIST=1 without a launcher: a one-rank group on this host
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(_free_port()))):
            os.environ.setdefault(k, v)
    if backend is None:
        import torch
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl":
        import torch
        torch.cuda.set_device(local_rank)
        kwargs["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend=backend, **kwargs)
    return dist


def max_over_ranks(dist, values, device="cpu"):
    """Element-wise MAX of a list of floats over all ranks (identity when dist is None)."""
    if dist is None:
        return list(values)
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def gather_ints(dist, value, device="cpu"):
    """All ranks' 63-bit integers, ordered by rank (e.g. per-shard checksums for the scaling report)."""
    if dist is None:
        return [int(value)]
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [int(x[0]) for x in out]
