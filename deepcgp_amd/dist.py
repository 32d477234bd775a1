"""Multi-GPU sharding of the forward ELBO (one process per GPU).

The data term of the ELBO is a sum over minibatch images and every layer's conditional is independent
per image, so the N images of a minibatch (each with all S samples and all P patches) are partitioned
contiguously over the ranks.  Parameters, Kuu, the Choleskys and the KL terms are replicated.  The only
exchange per step is a sum all-reduce of ONE float64 (the per-rank data term): an RCCL ``ncclAllReduce``
issued on the ctx stream inside ``dcgp_elbo_forward``.

Host-side plumbing (who am I, hand the 128-byte RCCL id to the other ranks, start/stop the clock together)
needs no framework: ``HostGroup`` is a few dozen lines of TCP -- rank 0 listens on MASTER_ADDR (loopback by
default), the others connect, every exchange is gather-to-0 + broadcast.  SINGLE NODE: the port rank 0 picked is
published through a file (mode 0600) that all ranks must see, which is what one node with 8 GPUs -- the
configuration this path is built for -- gives.  It is used for set-up and for the
timing barriers only, never on the data path.  ``spawn_ranks`` starts one process per GPU when the caller was
not already launched by a process launcher (RANK / WORLD_SIZE in the environment).
"""
import json
import os
import socket
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np


def shard_range(n, rank, world):
    """Contiguous partition of range(n); the first n % world ranks get one extra element."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world: %d/%d" % (rank, world))
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def grad_shard_range(n, world, rank):
    """Shards of a block of n values over `world` ranks as the sharded optimiser step cuts it (dcgp_shard_range, include/dcgp.h): every
    shard ceil(n / world) long -- RCCL's reduce-scatter / all-gather want equal counts, the block is padded to world * shard --, rank r
    holds [lo, hi) = [r * shard, min((r + 1) * shard, n)).  Returns (lo, hi, shard)."""
    if world <= 0 or not (0 <= rank < world) or n < 0:
        raise ValueError("bad n/world/rank: %d/%d/%d" % (n, world, rank))
    sh = -(-int(n) // int(world))
    return min(rank * sh, n), min((rank + 1) * sh, n), sh


def adam_update(p, g, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer's update of plain (untransformed) values ASCENDING the objective whose gradient is g, in place on (p, m, v):
    the arithmetic of opt_step_kernel (csrc/grad.hip) for transform-0 groups, for the host-side mirror of the sharded step."""
    gr = -g
    m *= beta1; m += (1.0 - beta1) * gr
    v *= beta2; v += (1.0 - beta2) * gr * gr
    p -= lr_t * m / (np.sqrt(v) + eps)


def sharded_adam_step(group, block_p, block_g_local, m, v, lr_t, **kw):
    """The multi-rank optimiser step of exchange mode 1 (dcgp_model_set_grad_exchange) on host arrays over a HostGroup: reduce-scatter of the
    rank-local gradient block, Adam on this rank's shard of the parameter block, all-gather of the parameters.  Returns the updated block
    (every rank the same).  The device runs the same three steps with ncclReduceScatter / opt_step_kernel / ncclAllGather."""
    n = block_p.size
    lo, hi, sh = grad_shard_range(n, group.world, group.rank)
    g_mine = group.reduce_scatter_sum(block_g_local)
    p_mine = block_p[lo:hi].copy()
    adam_update(p_mine, g_mine[:hi - lo], m[lo:hi], v[lo:hi], lr_t, **kw)
    stage = np.zeros(sh)
    stage[:hi - lo] = p_mine
    return group.all_gather(stage)[:n]


def shard_batch(X, Y, zs, rank, world):
    """This rank's images, labels and noise (z is indexed [S, image, D], so a shard sees exactly the
    rows it would see in the full batch -> results are independent of the number of ranks)."""
    lo, hi = shard_range(np.shape(X)[0], rank, world)
    zs_l = None if zs is None else [None if z is None else np.ascontiguousarray(z[:, lo:hi]) for z in zs]
    return X[lo:hi], Y[lo:hi], zs_l


def assemble_elbo(global_data_term, kl, num_data, global_batch):
    """ELBO = sum_n E_q log p(y_n) * num_data / batch - sum_l KL_l."""
    return global_data_term * (float(num_data) / float(global_batch)) - kl


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


# ---------------------------------------------------------------------------------------------------
# host-side process group over TCP
# ---------------------------------------------------------------------------------------------------
def _send_msg(sock, payload):
    sock.sendall(struct.pack("!I", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous socket")
        buf.extend(chunk)
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack("!I", _recv_exact(sock, 4))
    return _recv_exact(sock, n)


def rendezvous_file():
    """Where rank 0 publishes the port it listens on.  ``spawn_ranks`` hands every rank a fresh path in
    DCGP_RDZV_FILE; under an external launcher (torch.distributed.run, mpirun, ...) the ranks of one job share the
    launcher as parent process and the MASTER_PORT it chose, which names the file.  (MASTER_PORT itself belongs to
    the launcher's own store, so the group listens on an ephemeral port instead.)"""
    path = os.environ.get("DCGP_RDZV_FILE")
    if path:
        return path
    tag = "%s_%s_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "x"))
    return os.path.join(tempfile.gettempdir(), "dcgp_rdzv_" + "".join(c if c.isalnum() else "_" for c in tag))


class HostGroup:
    """Star-topology host process group: ``broadcast_bytes``, ``allreduce`` (small float vectors), ``barrier``."""

    def __init__(self, rank, world, addr=None, path=None, timeout=None):
        if timeout is None:
            timeout = float(os.environ.get("DCGP_HOSTGROUP_TIMEOUT", "120"))   # every wait of the group is bounded by it
        self.rank, self.world = int(rank), int(world)
        self._peers, self._sock, self._path = [], None, None
        if self.world <= 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        path = path or rendezvous_file()
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                srv.bind((addr, 0))        # the interface the other ranks connect to, not every interface
            except OSError:
                # MASTER_ADDR does not name a local interface (a hostname the container cannot resolve): the loopback, and the
                # address actually bound goes into the rendezvous file -- the peers connect to what the file says, so a single-node
                # job still meets; ranks on other nodes fail on the connect with the address in the message instead of timing out
                srv.bind(("127.0.0.1", 0))
            srv.listen(self.world)
            srv.settimeout(timeout)
            token = os.urandom(16).hex()
            tmp = path + ".%d.tmp" % os.getpid()
            try:
                os.unlink(tmp)             # left by a crashed earlier run whose pid has come round again
            except OSError:
                pass
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)   # the token is the group's only credential
            with os.fdopen(fd, "w") as fh:
                json.dump({"addr": srv.getsockname()[0], "port": srv.getsockname()[1], "token": token}, fh)
            os.replace(tmp, path)          # atomic: a reader sees the old file or the new one, never half of it
            self._path = path
            peers = {}
            while len(peers) < self.world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                hello = json.loads(_recv_msg(conn).decode())
                if hello.get("token") != token or not (0 < hello.get("rank", -1) < self.world) or hello["rank"] in peers:
                    conn.close()           # a stray process, or a rank of an earlier job that read a stale file
                    continue
                peers[hello["rank"]] = conn
                _send_msg(conn, b"ok")
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    with open(path) as fh:
                        info = json.load(fh)
                    s = socket.create_connection((info.get("addr") or addr, int(info["port"])), timeout=5.0)   # the address rank 0 really bound
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(timeout)
                    _send_msg(s, json.dumps({"rank": self.rank, "token": info["token"]}).encode())
                    if _recv_msg(s) == b"ok":
                        self._sock = s
                        break
                    s.close()
                except (OSError, ValueError, KeyError, ConnectionError):
                    pass                   # file not there yet / stale file of an earlier job: look again
                if time.time() > deadline:
                    raise TimeoutError("rank %d: no rendezvous with rank 0 through %s (MASTER_ADDR %s)" % (self.rank, path, addr))
                time.sleep(0.05)

    # every collective = gather to rank 0, combine there, send the result back
    def _exchange(self, payload, combine):
        if self.world <= 1:
            return combine([payload])
        if self.rank == 0:
            parts = [payload] + [_recv_msg(c) for c in self._peers]
            out = combine(parts)
            for c in self._peers:
                _send_msg(c, out)
            return out
        _send_msg(self._sock, payload)
        return _recv_msg(self._sock)

    def broadcast_bytes(self, b, src=0):
        if src != 0:
            raise ValueError("HostGroup broadcasts from rank 0")
        return self._exchange(bytes(b) if self.rank == 0 else b"", lambda parts: parts[0])

    def allreduce(self, values, op="sum"):
        """Elementwise sum / max / min of a short float vector over the ranks (fixed rank order: reproducible)."""
        fn = {"sum": np.sum, "max": np.max, "min": np.min}[op]
        v = np.ascontiguousarray(np.atleast_1d(values), np.float64)

        def combine(parts):
            return np.ascontiguousarray(fn(np.stack([np.frombuffer(p, np.float64) for p in parts]), axis=0)).tobytes()
        return np.frombuffer(self._exchange(v.tobytes(), combine), np.float64).copy()

    def reduce_scatter_sum(self, values):
        """Sum over the ranks of equally long float vectors, cut into `world` equal shards (the vector padded with zeros to world * shard,
        shard = ceil(n / world)): returns this rank's shard.  Fixed rank order: reproducible.  The host counterpart of ncclReduceScatter on a
        layer's gradient block (grad_shard_range)."""
        v = np.ascontiguousarray(np.atleast_1d(values), np.float64)
        sh = -(-v.size // max(self.world, 1))

        def combine(parts):
            tot = np.zeros(sh * max(self.world, 1))
            for p in parts:
                a = np.frombuffer(p, np.float64)
                if a.size != v.size:
                    raise ValueError("reduce_scatter_sum: ranks sent %d and %d values" % (v.size, a.size))
                tot[:a.size] += a
            return tot.tobytes()
        full = np.frombuffer(self._exchange(v.tobytes(), combine), np.float64)
        return full[self.rank * sh:(self.rank + 1) * sh].copy()

    def all_gather(self, shard):
        """Every rank's (equally long) float vector, concatenated in rank order, to every rank: the host counterpart of ncclAllGather."""
        v = np.ascontiguousarray(np.atleast_1d(shard), np.float64)

        def combine(parts):
            if any(len(p) != len(parts[0]) for p in parts):
                raise ValueError("all_gather: ranks sent shards of different lengths")
            return b"".join(parts)
        return np.frombuffer(self._exchange(v.tobytes(), combine), np.float64).copy()

    def barrier(self):
        self._exchange(b"", lambda parts: b"")

    def close(self):
        for c in self._peers:
            c.close()
        if self._sock is not None:
            self._sock.close()
        if self._path:
            try:
                os.unlink(self._path)
            except OSError:
                pass
        self._peers, self._sock, self._path = [], None, None


def spawn_ranks(n, argv=None, env=None):
    """Start ``n`` copies of this program, one per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR and a fresh
    DCGP_RDZV_FILE in their environment), wait for all of them and return the worst exit status.  Rank 0's stdout
    is this process's stdout."""
    argv = list(argv if argv is not None else sys.argv)
    fd, path = tempfile.mkstemp(prefix="dcgp_rdzv_")
    os.close(fd)
    os.unlink(path)
    procs = []
    for r in range(n):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), DCGP_RDZV_FILE=path)
        e.setdefault("MASTER_ADDR", "127.0.0.1")
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs across processes on this driver
        procs.append(subprocess.Popen([sys.executable] + argv, env=e, stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies takes the others with it (they would otherwise sit in a group exchange until the socket timeout)
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            if p.poll() is None:
                continue
            live.remove(p)
            if p.returncode and not rc:
                rc = p.returncode
                for q in live:
                    q.terminate()
    try:
        os.unlink(path)
    except OSError:
        pass
    return rc


def init_rccl(ctx, rank, world, broadcast_bytes):
    """Create the RCCL communicator of ``ctx``: rank 0 draws the unique id, ``broadcast_bytes(b)``
    (``HostGroup.broadcast_bytes``) ships the 128 bytes to the other ranks."""
    from . import device as dev
    uid = dev.comm_unique_id() if rank == 0 else bytes(128)
    uid = broadcast_bytes(uid)
    ctx.comm_init(world, rank, uid)
