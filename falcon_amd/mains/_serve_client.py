"""The job's side of falcon_amd.mains.consensus_server: hand this process's descriptors 0, 1, 2 and its command
line to the server behind $FALCON_AMD_SERVER and wait for the status.  Standard library only (this is all a job's
process does when a server is there).  Returns None whenever the job should run in its own process instead: no
variable, nobody listening, a request the server declines."""
import json
import os
import socket
import sys


def try_server(argv):
    path = os.environ.get("FALCON_AMD_SERVER")
    if not path:
        return None
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    try:
        s.settimeout(1.0)
        s.connect(path)
    except OSError:
        s.close()
        return None
    try:
        req = {"argv": list(argv), "skip_failed": bool(os.environ.get("FALCON_AMD_SKIP_FAILED_PILES"))}
        sys.stdout.flush()
        socket.send_fds(s, [json.dumps(req).encode()], [0, 1, 2])
        s.settimeout(None)
        buf = b""
        while True:
            chunk = s.recv(1 << 16)
            if not chunk:
                break
            buf += chunk
        if not buf:
            sys.stderr.write("falcon_amd: the consensus server went away while it had this job\n")
            return 1
        reply = json.loads(buf.decode())
    except (OSError, ValueError) as exc:
        sys.stderr.write("falcon_amd: the consensus server could not be talked to: %r\n" % (exc,))
        return 1
    finally:
        s.close()
    if reply.get("status") == "decline":
        return None
    if reply.get("message"):
        sys.stderr.write("falcon_amd: consensus server: %s\n" % reply["message"])
    return int(reply.get("status", 1))
