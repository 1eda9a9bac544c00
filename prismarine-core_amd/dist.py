"""Tile sharding of one frame across the GPUs of a node -- host plumbing only.

The reference has no multi-GPU path.  Radiance is per texel and no stage mixes texels, so a frame
shards by screen rows: rank r owns rows [y0,y1); the BVH is rebuilt redundantly on every GPU; the
per-texel frame radiance (sum rgb + deposit count, 16 B/texel) of every tile is gathered to rank 0
with ONE collective per frame (RCCL over xGMI: torch.distributed backend "nccl"), and rank 0 runs
the sampler over the whole image.  One tiny all-reduce per bounce round carries the global ray
count so the reference's `getRayCount() < 32 -> stop` rule (Pipeline.inl:459-461) is applied to the
whole frame: the sharded image equals the unsharded one.

torch is imported lazily and only when world > 1.
"""
import os


def tile_rows(rank, world, height):
    """Contiguous row strips of ceil(height/world) rows; returns (y0, y1, rows_per_rank)."""
    per = (height + world - 1) // world
    y0 = min(rank * per, height)
    return y0, min(y0 + per, height), per


class NativeUnavailable(RuntimeError):
    """The C ABI's RCCL communicator could not be created on some rank; raised on EVERY rank (Comm.attach_native
    agrees on it), so the caller can fall back to torch.distributed collectives as one."""


class NativeDist:
    """The C ABI's own communicator (psm_dist_*): the path's one data-path collective per frame (gather of the tiles'
    per-texel radiance to rank 0) and the small all-gather of (round, count) pairs, with no torch tensor in between.
    Two steps, as the C ABI has them: the constructor creates this rank's LOCAL resources (psm_dist_prepare: cannot
    block on a peer); connect_rccl / connect_hoststaged install the transport (collective).

    connect_rccl(bcast_id): RCCL directly from libpsm_hip.so; `bcast_id(bytes or None) -> bytes` hands rank 0's
    ncclUniqueId to the other ranks over whatever side channel the launcher has (bench.py: torch.distributed).
    connect_hoststaged(name, slot_bytes): the library's host-staged transport for processes that SHARE one GPU
    (tests; psm_dist_connect_hoststaged)."""

    def __init__(self, ctx, rank, world, bcast_id=None):
        import ctypes as C
        from . import lib
        self._C, self._lib, self.ctx, self.rank, self.world = C, lib(), ctx, rank, world
        self._h = C.c_void_p()
        ctx.check(self._lib.psm_dist_prepare(ctx._h, C.c_int(rank), C.c_int(world), C.byref(self._h)), "psm_dist_prepare")
        if bcast_id is not None:
            self.connect_rccl(bcast_id)

    def connect_rccl(self, bcast_id):
        C = self._C
        ident = (C.c_uint8 * 128)()
        rc = self._lib.psm_dist_unique_id(ident) if self.rank == 0 else 0
        # the id always travels (byte 128 = rank 0's return code), so a failure on rank 0 is every rank's failure
        # instead of a hang in the side channel
        raw = bcast_id(bytes(ident) + bytes([1 if rc else 0]) if self.rank == 0 else None)
        if raw[128]:
            raise NativeUnavailable("psm_dist_unique_id failed on rank 0 (rc %d)" % rc)
        ident = (C.c_uint8 * 128).from_buffer_copy(raw[:128])
        self.ctx.check(self._lib.psm_dist_connect(self._h, ident), "psm_dist_connect")

    def connect_hoststaged(self, shm_name, slot_bytes, timeout_ms=60000):
        C = self._C
        self.ctx.check(self._lib.psm_dist_connect_hoststaged(self._h, shm_name.encode(), C.c_size_t(slot_bytes), C.c_uint32(timeout_ms)),
                       "psm_dist_connect_hoststaged")

    @property
    def transport(self):
        self._lib.psm_dist_transport_name.restype = self._C.c_char_p
        v = self._lib.psm_dist_transport_name(self._h)
        return v.decode() if v else None

    @property
    def comm_ranks(self):
        """ranks the transport itself counts (RCCL: ncclCommCount); 0 while not connected"""
        return int(self._lib.psm_dist_comm_ranks(self._h))

    def agree(self, local_rc=0):
        """psm_dist_agree: 0 when every rank passed 0, this rank's code when it failed, PSM_ERR_PEER when only others did."""
        return self._lib.psm_dist_agree(self._h, self._C.c_int(local_rc))

    def gather_tiles(self, rays):
        """ONE collective: rays' owned texels -> rank 0's image (stream-ordered against rays' context, no host sync)."""
        rays.ctx.check(self._lib.psm_dist_gather_tiles(self._h, rays._h), "psm_dist_gather_tiles")

    def allgather_i32(self, values):
        C = self._C
        n = len(values)
        send = (C.c_int32 * n)(*[int(v) for v in values])
        recv = (C.c_int32 * (n * self.world))()
        self.ctx.check(self._lib.psm_dist_allgather_i32(self._h, send, recv, C.c_uint32(n)), "psm_dist_allgather_i32")
        return [list(recv[r * n:(r + 1) * n]) for r in range(self.world)]

    def barrier(self):
        self.ctx.check(self._lib.psm_dist_barrier(self._h), "psm_dist_barrier")

    def set_band_weights(self, weights):
        """the dealing of the bands the gathers use (psm_dist_set_band_weights); None = round-robin"""
        wv = None if weights is None else (self._C.c_uint32 * len(weights))(*[int(v) for v in weights])
        self.ctx.check(self._lib.psm_dist_set_band_weights(self._h, wv), "psm_dist_set_band_weights")

    def emulate_tile(self, tile_rank, tile_world):
        """one-GPU rehearsal of a worker's per-frame cost (psm_dist_emulate_tile)"""
        self.ctx.check(self._lib.psm_dist_emulate_tile(self._h, self._C.c_int(tile_rank), self._C.c_int(tile_world)), "psm_dist_emulate_tile")

    def close(self):
        if self._h:
            self._lib.psm_dist_destroy(self._h)
            self._h = None


class Comm:
    def __init__(self, world=1, backend=None, init=True, force=False):
        """force=True runs the distributed code path even for world == 1 (single-GPU rehearsal of the
        RCCL plumbing)."""
        self.world = world
        self.active = world > 1 or force
        self.rank = 0
        self.local_rank = 0
        self.device_index = 0
        self.backend = backend or os.environ.get("PSM_DIST_BACKEND", "nccl")
        self.torch = None
        if self.active:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            self.rank = int(os.environ.get("RANK", "0"))
            self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if self.backend == "nccl":
                ndev = torch.cuda.device_count()
                self.device_index = self.local_rank % max(ndev, 1)
                torch.cuda.set_device(self.device_index)
                self.dev = torch.device("cuda", self.device_index)
            else:
                ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
                self.device_index = self.local_rank % ndev if ndev else 0
                self.dev = torch.device("cpu")
            if init and not dist.is_initialized():
                dist.init_process_group(backend=self.backend, rank=self.rank, world_size=world)

    def attach_native(self, ctx, slot_bytes=0):
        """Create the C ABI's RCCL communicator on ctx's device (backend nccl only); rank 0's id travels by a
        torch.distributed broadcast. PSM_DIST_NATIVE=0 keeps every collective on torch.distributed.
        PSM_DIST_TRANSPORT=hoststaged (explicit, a rehearsal: ranks that SHARE a GPU, torch side channel over gloo)
        installs the library's host-staged transport instead of RCCL; slot_bytes = the largest tile of a gather."""
        self.native = None
        if not self.active or os.environ.get("PSM_DIST_NATIVE", "1") == "0":
            return None
        if os.environ.get("PSM_DIST_TRANSPORT", "rccl") == "hoststaged":
            err = None
            try:
                self.native = NativeDist(ctx, self.rank, self.world)
            except Exception as e:
                err = e
            if self.min_int(0 if err is not None else 1) == 0:
                self._drop_native()
                raise NativeUnavailable("psm_dist_prepare failed on %s: %s" % ("this rank" if err is not None else "another rank", err))
            # the segment's name carries a nonce of THIS run (rank 0's, summed over the side channel): a segment a killed run
            # left behind under the same MASTER_PORT can be neither in the way of rank 0's exclusive create nor be opened by
            # another rank in its place
            nonce = self.sum_int(int.from_bytes(os.urandom(6), "little") if self.rank == 0 else 0)
            try:
                self.native.connect_hoststaged("/psm-bench-%s-%012x" % (os.environ.get("MASTER_PORT", "0"), nonce), max(int(slot_bytes), 4096))
            except Exception as e:
                err = e
            if self.min_int(0 if err is not None else 1) == 0:
                self._drop_native()
                raise NativeUnavailable("psm_dist_connect_hoststaged failed on %s: %s" % ("this rank" if err is not None else "another rank", err))
            return self.native
        if self.backend != "nccl":
            return None

        def bcast(raw):
            t = self.torch.zeros(129, dtype=self.torch.uint8, device=self.dev)
            if raw is not None:
                t.copy_(self.torch.frombuffer(bytearray(raw), dtype=self.torch.uint8))
            if self.world > 1:
                self.dist.broadcast(t, src=0)
            return bytes(t.cpu().numpy().tobytes())
        # step 1, local: stream + events of the communicator. A rank that fails here says so BEFORE anybody enters the
        # collective ncclCommInitRank (where the others would wait for it for ever)
        err = None
        try:
            self.native = NativeDist(ctx, self.rank, self.world)
        except Exception as e:
            err = e
        if self.min_int(0 if err is not None else 1) == 0:
            self._drop_native()
            raise NativeUnavailable("psm_dist_prepare failed on %s: %s" % ("this rank" if err is not None else "another rank", err))
        # step 2, collective: rank 0's id travels, every rank joins the RCCL communicator
        try:
            self.native.connect_rccl(bcast)
        except NativeUnavailable:      # rank 0's failure, already known to every rank
            self._drop_native()
            raise
        except Exception as e:         # this rank's communicator failed: tell the others
            err = e
        if self.min_int(0 if err is not None else 1) == 0:
            self._drop_native()
            raise NativeUnavailable("psm_dist_connect failed on %s: %s" % ("this rank" if err is not None else "another rank", err))
        return self.native

    def _drop_native(self):
        if getattr(self, "native", None) is not None:
            self.native.close()
            self.native = None

    def min_int(self, v):
        if not self.active or self.world <= 1:
            return int(v)
        t = self.torch.tensor([int(v)], dtype=self.torch.int64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    def barrier(self):
        if self.active:
            self.dist.barrier()

    def sum_int(self, v):
        if not self.active:
            return int(v)
        t = self.torch.tensor([int(v)], dtype=self.torch.int64, device=self.dev)
        self.dist.all_reduce(t)
        return int(t.item())

    def max_float(self, v):
        if not self.active:
            return float(v)
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, values):
        """every rank's short list of floats, on every rank: [world][len(values)] (host path)"""
        if not self.active or self.world <= 1:
            return [[float(v) for v in values]]
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device=self.dev)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [[float(v) for v in o.cpu().tolist()] for o in out]

    def sum_ints(self, values):
        """Element-wise sum over ranks of a short list of ints (host path)."""
        if not self.active:
            return [int(v) for v in values]
        t = self.torch.tensor([int(v) for v in values], dtype=self.torch.int64, device=self.dev)
        self.dist.all_reduce(t)
        return [int(v) for v in t.cpu().tolist()]

    def gather_to_root(self, tile, root_out=None):
        """Gather equally sized 1-D tensors to rank 0. Returns the concatenated tensor on rank 0
        (root_out if given), None elsewhere. `tile` must live on self.dev."""
        if not self.active:
            return tile
        n = tile.numel()
        if self.rank == 0:
            if root_out is None:
                root_out = self.torch.empty(self.world * n, dtype=tile.dtype, device=tile.device)
            self.dist.gather(tile, list(root_out.split(n)), dst=0)
            return root_out
        self.dist.gather(tile, None, dst=0)
        return None

    def close(self):
        if getattr(self, "native", None) is not None:
            self.native.close()
            self.native = None
        if self.active and self.dist.is_initialized():
            try:
                self.dist.destroy_process_group()
            except RuntimeError:
                pass   # a peer that is already gone (every rank leaves together after a failed check: the slower one finds the socket closed)


def band_pattern(world, weights=None):
    """Owner of each position of one period of the band dealing (BandMap, csrc/psm_internal.h): global 8-row band g
    belongs to rank pattern[g % len(pattern)]. A smooth weighted round-robin: every step each rank's credit grows by its
    weight, the largest credit (lowest rank on ties) takes the band and pays the period. Weights all 1: 0, 1, .. world-1."""
    wts = [1] * world if weights is None else [int(v) for v in weights]
    assert len(wts) == world and 1 <= sum(wts) <= 64 and world <= 64
    period, cur, out = sum(wts), [0] * world, []
    for _ in range(period):
        pick = 0
        for r in range(world):
            cur[r] += wts[r]
            if cur[r] > cur[pick]:
                pick = r
        cur[pick] -= period
        out.append(pick)
    return out


def owned_texels(rank, world, width, height, weights=None):
    """Texels owned by `rank` when the 8-row bands are dealt by band_pattern (psm_rt_set_tile_interleaved / _weighted)."""
    pat = band_pattern(world, weights)
    rows = sum(min(8, height - 8 * g) for g in range((height + 7) // 8) if pat[g % len(pat)] == rank)
    return rows * width


def largest_tile_texels(world, width, height, weights=None):
    return max(owned_texels(r, world, width, height, weights) for r in range(world))


def interleaved_texels(rank, world, width, height):
    """Texels owned by `rank` when 8-row bands are dealt round-robin (psm_rt_set_tile_interleaved)."""
    return owned_texels(rank, world, width, height)


def default_band_weights(world):
    """The dealing bench.py uses on `world` GPUs: periods of 23 bands of which the gathering rank takes floor(23 / world)
    and the workers one more -- shares of 0.087 / 0.130 at 8 GPUs, 0.217 / 0.261 at 4, 0.478 / 0.522 at 2: what equalises
    the ranks' finishing times when rank 0's unpack + rest-of-image camera + whole-image sample cost 0.10 ms per 1080p
    frame against 2.36 ms of tracing per full image. None (round-robin) where that does not fit."""
    if world < 2 or world > 11:
        return None
    low = 23 // world
    rest = 23 - low
    if rest % (world - 1):
        return None
    return [low] + [rest // (world - 1)] * (world - 1)


def run_rounds(comm, rays, intersector, materials, depth=16, on_round=None):
    """The bounce loop of Viewer.cpp:304-310 for this rank's tile, in lock step with the other ranks:
    the `getRayCount() < 32 -> stop` rule is applied to the GLOBAL ray count.

    With backend nccl and a context created on torch's stream the per-round count exchange is
    device-side: copy the count to a tensor, all-gather it, ONE host read per round (the single-GPU
    loop also reads the count once per round). Otherwise (gloo / tests) counts go through the host."""
    from . import sharded_rounds
    device_exchange = (comm.active and comm.backend == "nccl" and getattr(comm, "same_stream", False))
    if not device_exchange:
        gen = sharded_rounds(rays, intersector, materials, depth)
        local = next(gen)
        rounds = 0
        while True:
            total = comm.sum_int(local)
            if on_round is not None and total >= 32:
                on_round(local)
            try:
                local = gen.send(total)
                rounds += 1
            except StopIteration:
                break
        return rounds
    torch, dist = comm.torch, comm.dist
    if not hasattr(comm, "_cnt"):
        comm._cnt = torch.zeros(1, dtype=torch.int32, device=comm.dev)
        comm._all = torch.zeros(comm.world, dtype=torch.int32, device=comm.dev)
    rays.applyMaterials(materials)
    local = rays.raycountCache
    total = comm.initial_total if getattr(comm, "initial_total", None) is not None else comm.sum_int(local)
    rounds = 0
    for _ in range(depth):
        if total < 32:
            break
        if on_round is not None:
            on_round(local)
        rays.intersection(intersector, force=True)
        rays.shade(force=True, reload=False)
        rays.reclaim()
        rays.ray_count_dev(comm._cnt.data_ptr())
        dist.all_gather_into_tensor(comm._all, comm._cnt)
        counts = comm._all.cpu()  # the one host synchronisation of the round
        local = int(counts[comm.rank])
        total = int(counts.sum())
        rays.set_ray_count(local)
        rounds += 1
    return rounds


def run_rounds_lanes(comm, lanes, depth=16, lane_streams=None, initial_totals=None):
    """The bounce loops of several frames in flight on this rank's tile (one frame per lane, each lane a
    (rays, intersector, materials) triple whose camera() has already run), in lock step with the other ranks:
    every round each live lane queues intersection + shade on its own stream, then ONE exchange carries all
    lanes' ray counts, and the `getRayCount() < 32 -> stop` rule is applied per frame to its GLOBAL count.
    Returns the number of rounds each frame ran.

    lane_streams (torch streams the lanes' contexts were created on, backend nccl): the counts stay on the
    device -- each lane copies its count into its slot of one tensor on its stream, the current stream waits
    for the lanes and all-gathers the tensor, one host read per round. Without it counts go through the host."""
    n = len(lanes)
    for rays, _, materials in lanes:
        rays.applyMaterials(materials)
    local = [rays.raycountCache for rays, _, _ in lanes]
    totals = list(initial_totals) if initial_totals is not None else comm.sum_ints(local)
    rounds = [0] * n
    device_exchange = lane_streams is not None and comm.active and comm.backend == "nccl"
    if device_exchange:
        torch, dist = comm.torch, comm.dist
        cnt = torch.zeros(n, dtype=torch.int32, device=comm.dev)
        allc = torch.zeros(comm.world * n, dtype=torch.int32, device=comm.dev)
        main = torch.cuda.current_stream()
    for _ in range(depth):
        live = [s for s in range(n) if totals[s] >= 32]
        if not live:
            break
        for s in live:
            rays, intersector, _ = lanes[s]
            rays.intersection(intersector, force=True)
            rays.shade(force=True, reload=False) if device_exchange else rays.shade(force=True)
            rays.reclaim()
            rounds[s] += 1
            if device_exchange:
                rays.ray_count_dev(cnt.data_ptr() + 4 * s)
        if device_exchange:
            for s in live:
                main.wait_stream(lane_streams[s])
            dist.all_gather_into_tensor(allc, cnt)
            counts = allc.cpu().view(comm.world, n)  # the one host synchronisation of the round
            for s in range(n):
                if s in live:
                    lanes[s][0].set_ray_count(int(counts[comm.rank, s]))
                    totals[s] = int(counts[:, s].sum())
                else:
                    totals[s] = 0
            for s in live:  # the next round's kernels must not overtake the exchange's read of cnt
                lane_streams[s].wait_stream(main)
        else:
            local = [lanes[s][0].raycountCache if s in live else 0 for s in range(n)]
            totals = comm.sum_ints(local)
    return rounds


def decide_sharded(all_rounds, all_counts, depth):
    """The stop rule of tile-sharded frames from what the ranks report once all their lanes are parked.
    all_rounds / all_counts: [world][lanes] (rounds done, local rays waiting). Returns per lane
    (over, force_until): a rank that is behind the furthest one catches up (any round below the furthest rank's
    had >= 32 rays there alone); with every rank at the same round the global count decides -- fewer than 32
    (or `depth` reached) ends the frame (Pipeline.inl:459-461), otherwise everybody runs one more round."""
    world, n = len(all_rounds), len(all_rounds[0])
    out = []
    for s in range(n):
        rs = [all_rounds[r][s] for r in range(world)]
        pmax = max(rs)
        if min(rs) < pmax:
            out.append((False, pmax))
        elif pmax >= depth or sum(all_counts[r][s] for r in range(world)) < 32:
            out.append((True, pmax))
        else:
            out.append((False, pmax + 1))
    return out


def run_batch_sharded(comm, batch, cam_inv, proj_inv, seeds, depth):
    """len(seeds) tile-sharded frames in flight on this rank (FrameBatch lanes, backend nccl). The lanes run free
    on the native scheduler until their local counts park them (psm_lanes_run_sharded); then ONE small
    all-gather tells every rank where the others stand and decide_sharded() applies the `< 32 rays -> stop`
    rule to each frame's global count; frames that go on are resumed. Returns rounds per frame."""
    torch, dist = comm.torch, comm.dist
    k = len(seeds)
    rounds, counts = batch.run_sharded(seeds, cam_inv, proj_inv, depth=depth)
    over = [False] * k
    native = getattr(comm, "native", None)
    while True:
        if native is not None:  # the C ABI's own all-gather (psm_dist_allgather_i32)
            flat = native.allgather_i32(list(rounds) + list(counts))
            allv = [[f[:k], f[k:]] for f in flat]
        else:
            mine = torch.tensor([rounds, counts], dtype=torch.int32).to(comm.dev)
            allv = torch.empty(comm.world * 2 * k, dtype=torch.int32, device=comm.dev)
            dist.all_gather_into_tensor(allv, mine.view(-1))
            allv = allv.cpu().view(comm.world, 2, k).tolist()
        verdict = decide_sharded([a[0] for a in allv], [a[1] for a in allv], depth)
        force = []
        for s in range(k):
            over[s] = over[s] or verdict[s][0]
            force.append(rounds[s] if over[s] else verdict[s][1])
        if all(over):
            return rounds
        # a finished frame's lane stays parked: its count is < 32 and force_until equals its round
        rounds, counts = batch.run_sharded(None, force_until=force, depth=depth)
