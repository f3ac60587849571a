"""Throughput of the resident-key daemon with and without its batching window (development aid): C concurrent
clients each send one 32-bit A+B (or A*B) over the socket, at the product parameter set.
DEVICES=0,0 (cloudd --devices): one evaluator per listed device, a round's jobs sliced across them; WINDOWS=100 limits the windows tried."""
import sys, os, time, tempfile, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ieache_amd as ia
from ieache_amd import daemon, tools

d = tempfile.mkdtemp(prefix="ieache_serve_")
tools.keygen_files(d)
clients = int(sys.argv[1]) if len(sys.argv) > 1 else 64
devices = [int(x) for x in os.environ["DEVICES"].split(",")] if os.environ.get("DEVICES") else None
windows = [int(x) for x in os.environ.get("WINDOWS", "0,100").split(",")]
for operator, name, f in ((1, "32-bit A+B", lambda a, b: a + b), (4, "32-bit A*B", lambda a, b: a * b)):
    blobs = []
    for i in range(clients):
        sub = os.path.join(d, "c%d_%d" % (operator, i))
        os.makedirs(sub)
        for k in ("secret.key", "nbit.key"):
            os.link(os.path.join(d, k), os.path.join(sub, k))
        tools.alice(sub, 0, 32, 1000 + i, seed=10 + i)
        tools.alice(sub, 0, 32, 7 * i + 1, seed=500 + i, append=True)
        blobs.append((sub, open(os.path.join(sub, "cloud.data"), "rb").read()))
    for window in windows:
        sock = os.path.join(d, "s%d_%d.sock" % (operator, window))
        proc = daemon.spawn(sock, os.path.join(d, "cloud.key"), batch_window_ms=window, max_batch=256, devices=devices)
        daemon.run_data(sock, operator, blobs[0][1])  # warm-up: circuit build, first launches
        answers = [None] * clients
        go = threading.Barrier(clients + 1)

        def client(i):
            go.wait()
            answers[i] = daemon.run_data(sock, operator, blobs[i][1])

        ts = [threading.Thread(target=client, args=(i,)) for i in range(clients)]
        for t in ts:
            t.start()
        go.wait()
        t0 = time.perf_counter()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        for i, (sub, _) in enumerate(blobs):
            rc, log, ans = answers[i]
            assert rc == 0, log
            open(os.path.join(sub, "answer.data"), "wb").write(ans)
            assert tools.verif_interpret(operator, *tools.verif(sub)) == f(1000 + i, 7 * i + 1)
        st = daemon.stats(sock)
        print("%s, %d concurrent clients, batching window %3d ms, devices %s: %.2f s for all (%.1f expressions/s); %s"
              % (name, clients, window, devices or [0], dt, clients / dt, st), flush=True)
        daemon.shutdown(sock)
        proc.wait(timeout=60)
