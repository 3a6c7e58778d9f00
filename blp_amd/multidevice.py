"""One process, one Python thread per device: the reference's own process model for several GPUs
(nn.DataParallel, /root/reference/train.py:329-330,344 -- SURVEY.md 8b "one Python thread per GPU
replica") applied to the evaluation, which the reference leaves on device 0 (train.py:79-80).

    group = DeviceGroup(["cuda:0", "cuda:1", ...])
    results = group.run(fn)              # fn(member) on one thread per device; results in rank order

A ``Member`` is what blp_amd.ranking takes as ``group``: rank / world plus the two collectives of an
evaluation -- ``all_reduce`` (sum, in place) and ``all_gather_into``.  Between distinct HIP
devices they are RCCL calls issued from ONE thread for all devices (torch.cuda.nccl: single-process group
launch, what DataParallel's replicate / reduce_add use); the threads meet at a barrier before and after.
Devices that repeat (["cuda:0", "cuda:0"]: two shards on the one GPU of a test box) or CPU tensors take
peer copies + a sum instead -- RCCL refuses two ranks on one device.

The C-ABI is re-entrant with an explicit (device, stream) per call and ctypes releases the GIL around it,
so the threads' ranking passes overlap; every thread works on its own stream.
"""
import contextlib
import threading

import torch


class DeviceThreadError(RuntimeError):
    """Another device thread of the group failed (its exception is the one DeviceGroup.run raises)."""


class Member:
    """One device thread's view of its DeviceGroup (rank, world, device, stream) + the collectives."""

    def __init__(self, group, rank):
        self.group = group
        self.rank = rank
        self.world = group.world
        self.device = group.devices[rank]
        self.stream = None  # the thread's own stream on a HIP device, made by DeviceGroup.run on the thread

    # ---------------------------------------------------------------- collectives
    def all_reduce(self, tensor):
        """Sum ``tensor`` over the group, in place (every member passes a tensor of the same shape / dtype)."""
        self.group._collective(self, "all_reduce", tensor)

    def all_gather_into(self, full, part):
        """full = [part of rank 0 | part of rank 1 | ...] on every member (flat, contiguous; full.numel() = world * part.numel())."""
        if full.numel() != self.world * part.numel() or not full.is_contiguous() or not part.is_contiguous():
            raise ValueError("all_gather_into: full must be contiguous with world * part.numel() elements")
        self.group._collective(self, "all_gather", (full.view(-1), part.view(-1)))

    def barrier(self):
        self.group._collective(self, "barrier", None)


class DeviceGroup:
    def __init__(self, devices, exchange="auto"):
        """devices: one entry per shard ("cuda:1", 1, torch.device, "cpu"); repeats allowed.
        exchange: "nccl" (RCCL group launch from one thread), "copy" (peer copies + sum), "auto" = nccl when every
        entry is a distinct HIP device and there is more than one."""
        self.devices = [self._device(d) for d in devices]
        if not self.devices:
            raise ValueError("DeviceGroup needs at least one device")
        self.world = len(self.devices)
        distinct = all(d.type == "cuda" for d in self.devices) and len({d.index for d in self.devices}) == self.world
        if exchange == "auto":
            exchange = "nccl" if distinct and self.world > 1 else "copy"
        if exchange not in ("nccl", "copy"):
            raise ValueError(f"unknown exchange {exchange!r}")
        if exchange == "nccl" and not distinct:
            raise ValueError("exchange='nccl' needs distinct HIP devices (RCCL refuses two ranks on one device)")
        self.exchange = exchange
        self.members = [Member(self, r) for r in range(self.world)]
        self.issued = []  # (op, bytes one member hands in), in issue order: what tests and bench's exchange log read
        self._barrier = threading.Barrier(self.world)
        self._slots = [None] * self.world
        self._failure = None
        self.nccl_error = None  # set when an RCCL group launch failed and the group fell back to peer copies (said in a warning)

    @staticmethod
    def _device(d):
        d = torch.device("cuda", d) if isinstance(d, int) else torch.device(d)
        if d.type == "cuda" and d.index is None:
            d = torch.device("cuda", torch.cuda.current_device())
        return d

    # ---------------------------------------------------------------- threads
    def run(self, fn):
        """fn(member) on one thread per member (member 0 on the calling thread, as DataParallel does not -- it saves a thread
        and keeps tracebacks of the common failure short); the results in rank order.  The first exception of any thread is
        re-raised here after every thread has stopped."""
        results, errors = [None] * self.world, [None] * self.world
        self._failure = None
        self._barrier.reset()

        def body(member):
            try:
                with self._on(member):
                    results[member.rank] = fn(member)
            except DeviceThreadError:
                pass  # a victim of another thread's failure
            except BaseException as exc:  # noqa: BLE001 -- carried to the caller
                errors[member.rank] = exc
                self._barrier.abort()  # nobody waits for this thread any more

        threads = [threading.Thread(target=body, args=(m,), name=f"blp-device-{m.rank}", daemon=True) for m in self.members[1:]]
        for t in threads:
            t.start()
        body(self.members[0])
        for t in threads:
            t.join()
        for exc in errors:
            if exc is not None:
                raise exc
        return results

    @contextlib.contextmanager
    def _on(self, member):
        if member.device.type != "cuda":
            yield
            return
        with torch.cuda.device(member.device):
            member.stream = torch.cuda.Stream(member.device)
            member.stream.wait_stream(torch.cuda.current_stream(member.device))  # what the caller queued comes first
            with torch.cuda.stream(member.stream):
                yield
            member.stream.synchronize()

    # ---------------------------------------------------------------- the meeting point
    def _wait(self):
        try:
            self._barrier.wait()
        except threading.BrokenBarrierError:
            raise DeviceThreadError("another device thread of the group failed") from None

    def _collective(self, member, op, payload):
        if self.world == 1:
            if op == "all_gather":
                payload[0].copy_(payload[1])
            return
        if member.stream is not None:
            member.stream.synchronize()  # this member's operand is complete before anyone touches it
        self._slots[member.rank] = payload
        self._wait()
        if member.rank == 0:
            try:
                self._issue(op)
            except BaseException as exc:  # noqa: BLE001
                self._failure = exc
        self._wait()
        if self._failure is not None:
            if member.rank == 0:
                raise self._failure
            raise DeviceThreadError("the collective failed on the issuing thread")

    def _issue_nccl(self, op, slots):
        from torch.cuda import nccl
        streams = [m.stream for m in self.members]
        if op == "all_reduce":
            nccl.all_reduce(list(slots), streams=streams)  # in place
        else:
            nccl.all_gather([part for _, part in slots], [full for full, _ in slots], streams=streams)
        for s in streams:
            s.synchronize()

    def _issue(self, op):
        """Member 0's thread, everybody else parked at the barrier: the collective for all members, then a synchronise of
        every device involved -- the members continue on their own streams with the result complete."""
        slots = self._slots
        if op == "barrier":
            return
        first = slots[0][1] if op == "all_gather" else slots[0]
        self.issued.append((op, first.numel() * first.element_size()))
        on_gpu = first.is_cuda
        if self.exchange == "nccl":
            try:
                self._issue_nccl(op, slots)
                return
            except Exception as exc:  # noqa: BLE001 -- RCCL's single-process group launch is not available here
                # Same numbers either way (sums of int32 counts, x + 0 = x for the vectors): the exchange changes, not a result.
                # Said out loud, once; the group stays on peer copies from here on.
                import warnings
                self.exchange, self.nccl_error = "copy", f"{type(exc).__name__}: {exc}"
                warnings.warn(f"blp_amd.multidevice: RCCL group launch failed ({self.nccl_error}); exchanging by peer copies instead")
        if op == "all_reduce":
            home = slots[0].device
            total = slots[0].clone()
            for t in slots[1:]:
                total += t.to(home)
            for t in slots:
                t.copy_(total)
        else:
            n = slots[0][1].numel()
            for full, _ in slots:
                for j, (_, part) in enumerate(slots):
                    full[j * n:(j + 1) * n].copy_(part)
        if on_gpu:
            for index in {d.index for d in self.devices if d.type == "cuda"}:
                torch.cuda.synchronize(index)
