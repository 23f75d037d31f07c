"""Rendezvous ports for the multi-process tests."""


def rendezvous_port() -> int:
    """A TCP port for a rendezvous on 127.0.0.1, from BELOW the kernel's ephemeral range (32768+): a port the kernel hands out for
    bind(0) can be taken again -- by some process's outgoing connection -- between closing the probe socket and the store's listen()
    (seen once: EADDRINUSE in a 2-rank test).  Ports here are only ever taken by explicit binds; each candidate is checked by binding it."""
    import os
    import random
    import socket
    rnd = random.Random(os.getpid() * 1000003 + int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        port = rnd.randrange(15000, 30000)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            try:
                sk.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:      # (last resort: the kernel's choice)
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
