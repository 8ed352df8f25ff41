"""Row tiling of one frame across the ranks of a node and the single gather that reassembles it.

The reference parallelises Renderer::render over rows with one RNG stream per row (raytrace.rs:176-185); here the
rows are dealt to ranks (one process per GPU) round-robin - rows differ in cost, so contiguous bands would unbalance - each rank renders its rows, and ONE gather (torch.distributed: RCCL over xGMI with backend "nccl", gloo
on CPU in the tests) brings the finished 8-bit rows to rank 0.  Because every sample's Philox stream is keyed by
(seed, pixel, sample), the assembled image is bit-identical for any world size.
"""
ROW_BLOCK = 1  # pure interleaving: every rank gets floor/ceil(H / world) rows (8-row blocks leave 800 rows over 8 ranks at 104 vs 96)


def rows_for_rank(height, rank, world_size, block=ROW_BLOCK):
    """Interleaved row blocks: block b of `block` rows goes to rank b % world_size."""
    return [j for j in range(height) if (j // block) % world_size == rank]


def max_rows_per_rank(height, world_size, block=ROW_BLOCK):
    return max(len(rows_for_rank(height, r, world_size, block)) for r in range(world_size))


def gather_rows(local_rows_u8, height, rank, world_size, block=ROW_BLOCK, group=None):
    """local_rows_u8: uint8 tensor [n_local_rows][W][3] holding this rank's rows in rows_for_rank order.
    Returns the assembled [H][W][3] uint8 tensor on rank 0 (None elsewhere).  One collective per frame."""
    import torch
    import torch.distributed as dist

    width = local_rows_u8.shape[1]
    if world_size == 1:
        image = torch.zeros((height, width, 3), dtype=torch.uint8, device=local_rows_u8.device)
        image[torch.tensor(rows_for_rank(height, 0, 1, block), device=local_rows_u8.device)] = local_rows_u8
        return image
    n_max = max_rows_per_rank(height, world_size, block)
    send = torch.zeros((n_max, width, 3), dtype=torch.uint8, device=local_rows_u8.device)
    send[: local_rows_u8.shape[0]] = local_rows_u8
    gathered = [torch.zeros_like(send) for _ in range(world_size)] if rank == 0 else None
    dist.gather(send, gathered, dst=0, group=group)
    if rank != 0:
        return None
    image = torch.zeros((height, width, 3), dtype=torch.uint8, device=local_rows_u8.device)
    for r in range(world_size):
        rr = rows_for_rank(height, r, world_size, block)
        image[torch.tensor(rr, device=local_rows_u8.device)] = gathered[r][: len(rr)]
    return image
