"""K-sharding over GPUs: one process per GPU, rollouts split into contiguous slices, ONE exchange of a
(T*C + 4)-float record per system and iteration.

Two drivers exist for the exchange:
  * inside the library: mppi_comm_init_rccl + ncclAllGather on the handle's stream (bench.py uses this);
  * outside, with torch.distributed: the classes below all-gather the handle's device buffers (mppi_get_exchange_buffers)
    in place — backend "nccl" (= RCCL over xGMI) on GPUs, and the same code runs over "gloo" with host tensors, which is
    how the multi-rank protocol is tested on machines without a GPU (tests/test_distributed_gloo.py).

Record layout (csrc/rollout_kernel.hpp partialStride): [U (T*C floats) | rho | eta | sum w^2 | pad]; every rank ends
with all records and merges them on the device (combineKernel), so every GPU holds the full u* (reference data flow for
comparison: three blocking D2H copies and two host scans per iteration, controllers/MPPI/mppi_controller.cu:187-219).
"""
import numpy as np


def shard_bounds(num_rollouts, rank, world_size):
    """(offset, count) of the rollouts rank owns: contiguous slices [r*K/G, (r+1)*K/G); K must divide evenly, like
    mppi_create requires (the special-trajectory rules and the Philox counters use the GLOBAL index offset + i)"""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    if num_rollouts % world_size != 0:
        raise ValueError("num_rollouts %d is not divisible by world_size %d" % (num_rollouts, world_size))
    k = num_rollouts // world_size
    return rank * k, k


def record_floats(num_timesteps, control_dim, num_systems=1):
    """floats one rank contributes per iteration: num_systems * (T*C + 4)"""
    return num_systems * (num_timesteps * control_dim + 4)


class RecordExchange:
    """all-gather of one fixed-size fp32 record per rank over a torch.distributed process group"""

    def __init__(self, floats_per_rank, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.n = int(floats_per_rank)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_gather(self, send, recv=None):
        """send: float32 tensor of n elements (host tensor for gloo, device tensor for nccl);
        returns recv [world][n] (allocated next to send unless given: the engine's own recv buffer on the GPU path)"""
        import torch
        assert send.dtype == torch.float32 and send.numel() == self.n
        if recv is None:
            recv = torch.empty((self.world, self.n), dtype=torch.float32, device=send.device)
        assert recv.numel() == self.world * self.n
        flat = recv.view(-1)
        try:
            self.dist.all_gather_into_tensor(flat, send.contiguous().view(-1), group=self.group)
        except (RuntimeError, NotImplementedError):  # backends without the flat variant
            parts = [flat[i * self.n:(i + 1) * self.n] for i in range(self.world)]
            self.dist.all_gather(parts, send.contiguous().view(-1), group=self.group)
        return recv.view(self.world, self.n)


def hip_runtimes_in_process():
    """distinct libamdhip64 files mapped into this process (PyTorch wheels bundle their own ROCm stack; device pointers
    and streams of one HIP runtime must not be handed to libraries bound to another)"""
    paths = set()
    try:
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                paths.add(line.split()[-1])
    except OSError:
        pass
    return sorted(paths)


class HostStagedExchange:
    """The same protocol with the records staged through host memory (mppi_read_send_record /
    mppi_write_recv_records) and gathered by any torch.distributed backend (gloo included).  Slower (two small copies
    and a stream sync per iteration) but independent of which ROCm runtime the collective library is bound to."""

    def __init__(self, controller, group=None):
        self.ctrl = controller
        self.exchange = RecordExchange(controller.exchangeBuffers()[2], group)

    def iterate(self, num_iterations=1):
        import torch
        for _ in range(num_iterations):
            self.ctrl.iterationLocal()
            rec = torch.from_numpy(self.ctrl.readSendRecord())
            self.ctrl.writeRecvRecords(self.exchange.all_gather(rec).numpy())
            self.ctrl.iterationMerge()


class _DeviceSpan:
    """exposes a raw device pointer through __cuda_array_interface__ so that torch can alias it without a copy"""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (int(nfloats),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class ShardedController:
    """Drives a controller created with (rank, world_size) through iterations whose exchange runs on torch.distributed.

    Kernels and the collective must be ordered on ONE stream: create a torch side stream, hand its raw handle to the
    controller, and this class issues the all-gather under that stream:
        s = torch.cuda.Stream()
        ctrl = VanillaMPPIController(..., rank=r, world_size=G, stream=s.cuda_stream)
        sharded = ShardedController(ctrl, s)
    (torch's default stream is the NULL stream, which the engine treats as "create my own" — do not pass that.)"""

    def __init__(self, controller, stream, group=None):
        import torch
        if not stream.cuda_stream:
            raise ValueError("ShardedController needs a non-default torch.cuda.Stream shared with the controller")
        rts = hip_runtimes_in_process()
        if len(rts) > 1:
            raise RuntimeError("two HIP runtimes are loaded (%s): torch cannot operate on this library's device buffers "
                               "and stream; use the library's own RCCL driver (mppi_comm_init_rccl) or HostStagedExchange"
                               % ", ".join(rts))
        self.ctrl = controller
        self.stream = stream
        send, recv, n = controller.exchangeBuffers()
        self.exchange = RecordExchange(n, group)
        self.send = torch.as_tensor(_DeviceSpan(send, n), device="cuda")
        self.recv = torch.as_tensor(_DeviceSpan(recv, n * self.exchange.world), device="cuda")

    def iterate(self, num_iterations=1):
        import torch
        with torch.cuda.stream(self.stream):
            for _ in range(num_iterations):
                self.ctrl.iterationLocal()
                self.exchange.all_gather(self.send, self.recv)  # a world of one copies send -> recv (force_exchange)
                self.ctrl.iterationMerge()


def merge_rule_float64(U, rho, eta, lambda_):
    """The merge combineKernel performs, restated in float64 for host-side checks of the protocol (NOT used on the
    product path — the device merges): rho = min rho_g, s_g = exp(-(rho_g - rho)/lambda), eta = sum s_g eta_g,
    u* = sum s_g U_g / eta."""
    U, rho, eta = np.asarray(U, np.float64), np.asarray(rho, np.float64), np.asarray(eta, np.float64)
    rho_min = rho.min()
    s = np.exp(-(rho - rho_min) / lambda_)
    eta_tot = float((s * eta).sum())
    return (s[:, None] * U).sum(0) / eta_tot, float(rho_min), eta_tot
