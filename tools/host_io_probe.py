"""Where does the PCIe-inclusive leg stop?  H2D alone, D2H alone, both at once, for the byte counts of the BGR (3 B/px) and the YUYV (2 B/px) step at 256 x 640x480,
through pinned buffers on separate streams — and with the copies cut into pieces (several copies in flight per direction).  Run on the GPU box."""
import time
import torch

def rate(nbytes, fn, iters=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return nbytes * iters / (time.perf_counter() - t) / 1e9

for label, nb in (("yuyv 256 x VGA x 2 B", 256 * 640 * 480 * 2), ("bgr 256 x VGA x 3 B", 256 * 640 * 480 * 3)):
    h_in = torch.empty(nb, dtype=torch.uint8).pin_memory(); h_out = torch.empty(nb, dtype=torch.uint8).pin_memory()
    d_a = torch.empty(nb, dtype=torch.uint8, device="cuda"); d_b = torch.empty(nb, dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def h2d():
        with torch.cuda.stream(s1): d_a.copy_(h_in, non_blocking=True)
    def d2h():
        with torch.cuda.stream(s2): h_out.copy_(d_b, non_blocking=True)
    def both():
        h2d(); d2h()
    print("%-22s H2D alone %5.1f GB/s   D2H alone %5.1f GB/s   both: %5.1f GB/s each way" % (label, rate(nb, h2d), rate(nb, d2h), rate(nb, both)))
    for pieces in (2, 4, 8):
        sz = nb // pieces
        ss1 = [torch.cuda.Stream() for _ in range(pieces)]; ss2 = [torch.cuda.Stream() for _ in range(pieces)]
        def both_p():
            for k in range(pieces):
                with torch.cuda.stream(ss1[k]): d_a[k * sz:(k + 1) * sz].copy_(h_in[k * sz:(k + 1) * sz], non_blocking=True)
                with torch.cuda.stream(ss2[k]): h_out[k * sz:(k + 1) * sz].copy_(d_b[k * sz:(k + 1) * sz], non_blocking=True)
        print("   %d pieces per direction on their own streams: %5.1f GB/s each way" % (pieces, rate(nb, both_p)))
