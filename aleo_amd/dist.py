"""Point-sharded MSM across the GPUs of one node (BASELINE.json config 5).

An MSM is a sum over (scalar, base) pairs, so rank r owns the contiguous shard [r*n/G, (r+1)*n/G): its bases stay
resident in its HBM, it runs the whole Pippenger locally and produces ONE partial result (a Jacobian point, 144 B).
The only exchange is an all-gather of G x 144 bytes (RCCL over xGMI when the backend is "nccl"; an EC-point sum is
not expressible as an RCCL reduction op, so "all-reduce" = all-gather + local group add).  Every rank then adds the
G partials in rank order, so all ranks hold identical bytes.  No other collective is on the data path."""
from __future__ import annotations
import numpy as np


def shard_range(n: int, rank: int, world: int):
    """Contiguous, balanced shard of [0, n) for `rank` (first n % world ranks get one extra element)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_partials(partial: np.ndarray, group=None, device=None) -> np.ndarray:
    """partial: uint64[k] on the host (k = 18 for a Jacobian point) -> uint64[world,k] on every rank.
    With backend "nccl" (= RCCL) pass the rank's device: the 144-byte payload travels over xGMI."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint64).reshape(-1).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = torch.empty((world, t.numel()), dtype=t.dtype, device=t.device)
    try:
        dist.all_gather_into_tensor(out.view(-1), t, group=group)          # one collective, one read-back
    except (RuntimeError, NotImplementedError, AttributeError):
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        out = torch.stack(outs)
    return out.cpu().numpy().view(np.uint64)


class PartialGather:
    """The per-step exchange of the sharded MSM with its buffers allocated once: a 144-byte partial goes from the host tail
    into a resident device tensor, ONE all-gather (RCCL over xGMI for a device, gloo for host tensors), one read-back."""

    def __init__(self, world: int, device=None, group=None, words: int = 18):
        import torch
        self.world, self.group = world, group
        self.src = torch.zeros(words, dtype=torch.int64, device=device if device is not None else 'cpu')
        self.dst = torch.zeros((world, words), dtype=torch.int64, device=self.src.device)
        self.pin = torch.zeros(words, dtype=torch.int64).pin_memory() if device is not None else None

    def __call__(self, partial: np.ndarray) -> np.ndarray:
        import torch
        import torch.distributed as dist
        p = np.ascontiguousarray(partial, dtype=np.uint64).reshape(-1).view(np.int64)
        k = p.shape[0]
        if self.pin is not None:
            self.pin[:k].copy_(torch.from_numpy(p)); self.src[:k].copy_(self.pin[:k], non_blocking=True)
        else:
            self.src[:k].copy_(torch.from_numpy(p))
        dist.all_gather_into_tensor(self.dst.view(-1), self.src, group=self.group)
        return self.dst.cpu().numpy().view(np.uint64)[:, :k]


def sharded_msm(local_msm, combine, partial_for_rank=None, group=None, device=None) -> np.ndarray:
    """local_msm() -> uint64[18] partial of this rank's shard; combine(uint64[G,18]) -> uint64[18].
    The product passes VariableBase.msm over the rank's pinned shard and aleo_amd.g1_sum."""
    part = local_msm()
    allp = all_gather_partials(part, group=group, device=device)
    return combine(allp)


# ---------------------------------------------------------------------------------------------------------------------
# Sharded NTT (SURVEY.md §8e "NTT (if sharded): 4-step"; BASELINE.json north_star: "shard ... NTT coefficients across the
# 8 GPUs").  Domain size n = R * C.  With j = r*C + c and k = k_c*R + k_r,
#     X[k_c*R + k_r] = sum_c w_C^(c k_c) * w_n^(c k_r) * ( sum_r x[r*C + c] * w_R^(r k_r) ),
# so a forward transform is: length-R transforms down the columns, the twiddle w_n^(c k_r), ONE all-to-all that turns the
# column distribution into a row distribution (RCCL over xGMI: every rank sends one block to every peer, one peer per
# link), and length-C transforms along the rows.  Layouts (G ranks, Cg = C/G, Rg = R/G):
#     coefficient layout  rank g holds x[r*C + c] for c in [g*Cg, (g+1)*Cg), all r            as int64[R, Cg, 4]
#     evaluation layout   rank h holds X[k_c*R + k_r] for k_r in [h*Rg, (h+1)*Rg), all k_c    as int64[Rg, C, 4]
# Pointwise polynomial arithmetic is layout-agnostic, and the inverse transform maps the evaluation layout back to the
# coefficient layout, so a prover never needs the natural order on one rank.  Each local step is a call into
# libaleo_mi355x.so (aleo_mi355x_ntt_fr_batch_device / aleo_mi355x_fr_grid_scale_device); torch only transposes and
# exchanges.  There is no other collective on the data path.
def _torch_stream_handle() -> int:
    """torch's current stream for the library: its handle, or hipStreamLegacy (1) for the default stream — a NULL stream
    argument would mean "the library's own stream", which is not ordered with torch's copies and transposes."""
    import torch
    return torch.cuda.current_stream().cuda_stream or 1


class HipLocalOps:
    """The product's local steps: HIP kernels on device-resident int64[..., 4] tensors (Montgomery Fr limbs), queued on
    torch's current stream so they are ordered with the transposes and the exchange around them."""

    def batch_ntt(self, t, lg_len: int, batch: int, direction: int):
        import ctypes
        import torch
        from ._lib import lib, check
        assert t.is_cuda and t.is_contiguous()
        check(lib().aleo_mi355x_ntt_fr_batch_device(ctypes.c_void_p(t.data_ptr()), lg_len, batch, 0, direction, 0,
                                                    ctypes.c_void_p(_torch_stream_handle())), 'ntt_fr_batch_device')

    def grid_scale(self, t, lg_n: int, rows: int, cols: int, row0: int, col0: int, ld: int, mode: int, direction: int):
        import ctypes
        import torch
        from ._lib import lib, check
        assert t.is_cuda and t.is_contiguous()
        check(lib().aleo_mi355x_fr_grid_scale_device(ctypes.c_void_p(t.data_ptr()), lg_n, rows, cols, row0, col0, ld, mode, direction,
                                                     ctypes.c_void_p(_torch_stream_handle())), 'fr_grid_scale_device')


class ShardedDomain:
    """EvaluationDomain<Fr> of size n = 2^lg_n spread over the ranks of `group` (see the layout note above).
    `exchange(blocks) -> blocks` performs the all-to-all (default: torch.distributed.all_to_all_single on `group`);
    `ops` performs the local steps (default: the HIP library)."""

    def __init__(self, lg_n: int, rank: int, world: int, ops=None, group=None, lg_rows: int = None, exchange=None, always_collective: bool = False):
        if world & (world - 1): raise ValueError('world size must be a power of two')
        lg_g = world.bit_length() - 1
        self.lg_n, self.rank, self.world, self.group = lg_n, rank, world, group
        self.lg_r = lg_rows if lg_rows is not None else lg_n // 2
        self.lg_c = lg_n - self.lg_r
        if self.lg_r < lg_g or self.lg_c < lg_g: raise ValueError('domain too small for this many ranks')
        self.R, self.C = 1 << self.lg_r, 1 << self.lg_c
        self.Rg, self.Cg = self.R // world, self.C // world
        self.ops = ops if ops is not None else HipLocalOps()
        self.exchange = exchange                 # tests: exchange(send[world, ...], rank) -> recv, instead of torch.distributed
        self.always_collective = always_collective      # a world of one rank still calls torch.distributed.all_to_all_single (how a one-GPU box runs the RCCL launch)
        self.collective_calls = 0                # all_to_all_single calls made so far

    # -- layouts (host helpers for tests / loading) --------------------------------------------------------------
    def coefficient_shard(self, x_full: np.ndarray) -> np.ndarray:
        """natural-order uint64[n,4] -> this rank's int64-viewable [R, Cg, 4] block."""
        m = np.ascontiguousarray(x_full, dtype=np.uint64).reshape(self.R, self.C, 4)
        return np.ascontiguousarray(m[:, self.rank * self.Cg:(self.rank + 1) * self.Cg])

    def evaluation_indices(self) -> np.ndarray:
        """natural index k = k_c*R + k_r of every element of this rank's [Rg, C] evaluation block."""
        kr = np.arange(self.rank * self.Rg, (self.rank + 1) * self.Rg, dtype=np.int64)[:, None]
        kc = np.arange(self.C, dtype=np.int64)[None, :]
        return kc * self.R + kr

    # -- the exchange -------------------------------------------------------------------------------------------------
    def _all_to_all(self, send):
        """send: [world, ...] (block p goes to rank p) -> [world, ...] (block p came from rank p)."""
        import torch
        import torch.distributed as dist
        send = send.contiguous()
        if self.world == 1 and not self.always_collective: return send
        if self.exchange is not None: return self.exchange(send, self.rank)
        if send.is_cuda and dist.get_backend(self.group) == 'gloo':      # rehearsal backend: stage through the host
            h = send.cpu(); r = torch.empty_like(h)
            dist.all_to_all_single(r, h, group=self.group)
            return r.to(send.device)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)                # RCCL over xGMI
        self.collective_calls += 1
        return recv

    # -- transforms -----------------------------------------------------------------------------------------------------
    def forward(self, coeffs, coset: bool = False):
        """coefficient layout [R, Cg, 4] -> evaluation layout [Rg, C, 4] (new tensor; `coeffs` is consumed)."""
        G, R, C, Rg, Cg = self.world, self.R, self.C, self.Rg, self.Cg
        t = coeffs.contiguous()
        if coset: self.ops.grid_scale(t, self.lg_n, R, Cg, 0, self.rank * Cg, C, 1, 0)
        t = t.permute(1, 0, 2).contiguous()                                   # [Cg, R]: every column contiguous
        self.ops.batch_ntt(t, self.lg_r, Cg, 0)                               # [cl][k_r]
        self.ops.grid_scale(t, self.lg_n, Cg, R, self.rank * Cg, 0, 0, 0, 0)  # *= w_n^(c * k_r)
        send = t.reshape(Cg, G, Rg, 4).permute(1, 0, 2, 3)                    # block h = my columns, rank h's k_r range
        recv = self._all_to_all(send)                                         # [g][cl][k_rl] = all columns, my k_r range
        t = recv.reshape(C, Rg, 4).permute(1, 0, 2).contiguous()              # [k_rl][c]
        self.ops.batch_ntt(t, self.lg_c, Rg, 0)                               # [k_rl][k_c]
        return t

    def inverse(self, evals, coset: bool = False):
        """evaluation layout [Rg, C, 4] -> coefficient layout [R, Cg, 4] (new tensor; `evals` is consumed)."""
        G, R, C, Rg, Cg = self.world, self.R, self.C, self.Rg, self.Cg
        t = evals.contiguous()
        self.ops.batch_ntt(t, self.lg_c, Rg, 1)                               # [k_rl][c], scaled by C^-1
        self.ops.grid_scale(t, self.lg_n, Rg, C, self.rank * Rg, 0, 0, 0, 1)  # *= w_n^-(k_r * c)
        send = t.reshape(Rg, G, Cg, 4).permute(1, 0, 2, 3)                    # block g = my k_r range, rank g's columns
        recv = self._all_to_all(send)                                         # [h][k_rl][cl] = all k_r, my columns
        t = recv.reshape(R, Cg, 4).permute(1, 0, 2).contiguous()              # [cl][k_r]
        self.ops.batch_ntt(t, self.lg_r, Cg, 1)                               # [cl][r], scaled by R^-1
        t = t.permute(1, 0, 2).contiguous()                                   # [r][cl]
        if coset: self.ops.grid_scale(t, self.lg_n, R, Cg, 0, self.rank * Cg, C, 1, 1)
        return t
