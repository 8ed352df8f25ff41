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


class RowGatherer:
    """gather_rows with everything that does not change between frames built once: the send / receive buffers, the
    row-index tensors and the frame.  At 8 GPUs a frame of the headline config takes ~10 ms per rank, so per-frame
    allocations, list building and host-to-device index copies would be a visible fraction of the step."""

    def __init__(self, height, width, rank, world_size, device, block=ROW_BLOCK, group=None):
        import torch

        self.height, self.rank, self.world_size, self.group = height, rank, world_size, group
        self.my_rows = rows_for_rank(height, rank, world_size, block)
        n_max = max_rows_per_rank(height, world_size, block)
        self.send = torch.zeros((n_max, width, 3), dtype=torch.uint8, device=device)
        self.image = torch.zeros((height, width, 3), dtype=torch.uint8, device=device) if rank == 0 else None
        # rank 0 receives into ONE buffer (the per-rank tensors dist.gather wants are views of it), so that when every rank holds the
        # same number of rows the frame is assembled by one index_copy_ instead of one per rank
        self.flat = torch.zeros((world_size * n_max, width, 3), dtype=torch.uint8, device=device) if (rank == 0 and world_size > 1) else None
        self.gathered = list(self.flat.chunk(world_size, dim=0)) if self.flat is not None else None
        self.index = None
        self.flat_index = None
        if rank == 0:
            self.index = [torch.tensor(rows_for_rank(height, r, world_size, block), dtype=torch.long, device=device) for r in range(world_size)]
            if world_size > 1 and all(ix.numel() == n_max for ix in self.index):
                self.flat_index = torch.cat(self.index)

    def gather(self, local_rows_i32):
        """local_rows_i32: int32 [n_local][W][3] (values 0..255) on this rank's device.  Returns the frame on rank 0."""
        import torch.distributed as dist

        n = len(self.my_rows)
        self.send[:n].copy_(local_rows_i32)  # int32 -> uint8 conversion fused into the copy
        if self.world_size == 1:
            self.image.index_copy_(0, self.index[0], self.send[:n])
            return self.image
        dist.gather(self.send, self.gathered, dst=0, group=self.group)  # the one collective of the frame
        if self.rank != 0:
            return None
        if self.flat_index is not None:
            self.image.index_copy_(0, self.flat_index, self.flat)
            return self.image
        for r in range(self.world_size):
            self.image.index_copy_(0, self.index[r], self.gathered[r][: self.index[r].numel()])
        return self.image
