"""One process per GPU: how the ranks of a multi-GPU run find each other and exchange.

The data path is RCCL over xGMI INSIDE libpandora_amd.so (csrc/pmx_comm.hip: ncclAllReduce / ncclAllGather on device buffers the
context owns).  This module only bootstraps it - rank 0 draws the 128-byte RCCL id (pmx_comm_unique_id) and hands it to the other
ranks over a TCP socket at MASTER_ADDR (the launcher's rendezvous address; any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK /
MASTER_ADDR / MASTER_PORT works, `python -m torch.distributed.run` included).  RCCL is the ONLY transport of this module: there
is no environment switch and no host fallback.  The host-side stand-ins that let the exchange STEPS be exercised where RCCL cannot
run the ranks (two ranks on one GPU, or no GPU at all) are test infrastructure and live in tests/transports.py (subclasses of
Comm that a test constructs explicitly).  No PyTorch anywhere in this file."""
import os
import socket
import struct
import time

import numpy as np

ID_BYTES = 128
OPS = {"min": 0, "sum": 1, "max": 2}
_NP_OPS = {"min": np.minimum, "sum": np.add, "max": np.maximum}


def env_world():
    """(rank, world, local_rank, master_addr, master_port) from the launcher's environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")),
            os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")))


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous socket")
        buf += chunk
    return bytes(buf)


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class Rendezvous:
    """Star of TCP connections through rank 0 (listening at addr:port): enough to hand out the RCCL id, to run a barrier and, for
    the "tcp" test transport, to reduce host arrays.  Not a data path."""

    def __init__(self, rank, world, addr, port, timeout=120.0):
        self.rank, self.world = rank, world
        self.peers = []  # rank 0: sockets of ranks 1..world-1 (index = rank - 1); others: [socket to rank 0]
        if world == 1:
            return
        # the port may be taken by somebody else's service: rank 0 listens at the first free one of a short list of candidates,
        # the others try the candidates in turn and only stay where the greeting is answered by this run's rank 0
        ports = [port + k * 97 for k in range(8)]
        greeting = b"PMX-RDV1" + struct.pack("<I", world)
        if rank == 0:
            srv = None
            for cand in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, cand))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise OSError(f"rendezvous: none of the ports {ports} at {addr} is free")
            srv.listen(world)
            srv.settimeout(timeout)
            by_rank = {}
            while len(by_rank) < world - 1:
                conn, _ = srv.accept()
                conn.settimeout(timeout)
                try:
                    hello = _recv_exact(conn, len(greeting) + 4)
                except (OSError, ConnectionError):
                    conn.close()
                    continue
                if hello[:len(greeting)] != greeting:  # not one of ours
                    conn.close()
                    continue
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.sendall(greeting)
                (r,) = struct.unpack("<I", hello[len(greeting):])
                by_rank[r] = conn
            srv.close()
            self.peers = [by_rank[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            s = None
            while s is None:
                for cand in ports:
                    try:
                        c = socket.create_connection((addr, cand), timeout=5.0)
                        c.settimeout(5.0)
                        c.sendall(greeting + struct.pack("<I", rank))
                        if _recv_exact(c, len(greeting)) == greeting:
                            s = c
                            break
                        c.close()
                    except (OSError, ConnectionError):
                        pass
                if s is None:
                    if time.time() > deadline:
                        raise TimeoutError(f"rendezvous: rank 0 did not answer at {addr}, ports {ports}")
                    time.sleep(0.05)
            s.settimeout(timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self.peers = [s]

    def broadcast(self, payload=None):
        """rank 0's bytes to everyone"""
        if self.world == 1:
            return payload
        if self.rank == 0:
            for p in self.peers:
                _send_msg(p, payload)
            return payload
        return _recv_msg(self.peers[0])

    def allreduce(self, arr, op):
        """host array reduced over the ranks (test transport)"""
        if self.world == 1:
            return arr
        a = np.ascontiguousarray(arr)
        if self.rank == 0:
            acc = a.copy()
            for p in self.peers:
                other = np.frombuffer(_recv_msg(p), a.dtype).reshape(a.shape)
                acc = _NP_OPS[op](acc, other)
            raw = acc.tobytes()
            for p in self.peers:
                _send_msg(p, raw)
            return acc
        _send_msg(self.peers[0], a.tobytes())
        return np.frombuffer(_recv_msg(self.peers[0]), a.dtype).reshape(a.shape).copy()

    def allgather(self, payload):
        """list of every rank's bytes, in rank order"""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv_msg(p) for p in self.peers]
            blob = b"".join(struct.pack("<Q", len(x)) + x for x in parts)
            for p in self.peers:
                _send_msg(p, blob)
            return parts
        _send_msg(self.peers[0], payload)
        blob, parts, pos = _recv_msg(self.peers[0]), [], 0
        while pos < len(blob):
            (n,) = struct.unpack_from("<Q", blob, pos)
            parts.append(blob[pos + 8:pos + 8 + n])
            pos += 8 + n
        return parts

    def barrier(self):
        self.allreduce(np.zeros(1, np.int32), "sum")

    def close(self):
        for p in self.peers:
            try:
                p.close()
            except OSError:
                pass
        self.peers = []


class _StdoutToStderr:
    """RCCL prints a version banner on file descriptor 1 when a communicator is created; a caller whose stdout is a protocol
    (bench.py: ONE JSON line) gets it on stderr instead."""

    def __enter__(self):
        import sys

        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self.saved, 1)
        os.close(self.saved)


class Comm:
    """The ranks of one run: RCCL collectives on the engine's device exchange buffers, on the engine's stream."""

    def __init__(self, engine, rank=None, world=None, addr=None, port=None, always=False):
        erank, eworld, _, eaddr, eport = env_world()
        self.rank = erank if rank is None else rank
        self.world = eworld if world is None else world
        self.engine = engine
        self.always = always  # run the collectives even with one rank (tests of the RCCL path on a one-GPU box)
        self.rdv = None
        self._bootstrap(eaddr if addr is None else addr, eport if port is None else port)

    def _bootstrap(self, addr, port):
        if self.engine is None:
            raise ValueError("the RCCL transport works on an engine's device buffers")
        # the launcher's port belongs to the launcher (torch.distributed.run keeps its store there): rendezvous one above
        port = int(os.environ.get("PANDORA_COMM_PORT", port + 1))
        self.rdv = Rendezvous(self.rank, self.world, addr, port)
        with _StdoutToStderr():
            uid = self.rdv.broadcast(self.engine.comm_unique_id() if self.rank == 0 else None)
            self.engine.comm_init(uid, self.world, self.rank)

    @property
    def nranks(self):
        """what RCCL itself says (ncclCommCount): bench.py prints it next to WORLD_SIZE"""
        return self.engine.comm_count()

    # ---- host values ---------------------------------------------------------------------------------------------
    def host_allreduce(self, arr, op):
        """small host arrays (timings, test vectors): eight doubles at a time through the device."""
        if self.world == 1:
            return np.asarray(arr)
        a = np.asarray(arr, np.float64).ravel()
        out = np.empty_like(a)
        for i in range(0, a.size, 8):
            chunk = np.zeros(8, np.float64)
            chunk[:a[i:i + 8].size] = a[i:i + 8]
            red = self.engine.comm_allreduce_scalars(chunk, op)
            out[i:i + 8] = red[:a[i:i + 8].size]
        return out.reshape(np.shape(arr))

    def barrier(self):
        if self.world == 1:
            return
        self.engine.sync()
        self.host_allreduce(np.zeros(1), "sum")  # a collective on the stream + the download that waits for it

    # ---- device exchange buffers -------------------------------------------------------------------------------------
    def allreduce_xbuf(self, which, op):
        """In-place reduction of one of the engine's exchange buffers over the ranks."""
        if self.world == 1 and not self.always:
            return
        self.engine.comm_allreduce(which, op)

    def allgather_rows(self, H, with_itp):
        """Every rank placed its owned rows in the engine's full-size maps; afterwards every rank holds all rows."""
        if self.world == 1 and not self.always:
            return
        self.engine.comm_allgather_rows(with_itp)

    def gather_rows(self, H, with_itp, root=0):
        """Every rank placed its owned rows in the engine's full-size maps; afterwards rank `root` holds all rows."""
        if self.world == 1 and not self.always:
            return
        self.engine.comm_gather_rows(root, with_itp)

    def close(self):
        if self.engine is not None and self.rdv is not None:
            self.engine.comm_destroy()
        if self.rdv is not None:
            self.rdv.close()
