"""Host link of one GPU: page-locked H2D and D2H copy rates alone and at the same time (two
streams), at the sizes of the 500-frame front-end pass (104 MB up, 28 MB down)."""
import time, torch
up = torch.empty(104 << 20, dtype=torch.uint8, pin_memory=True)
down = torch.empty(28 << 20, dtype=torch.uint8, pin_memory=True)
d_up = torch.empty_like(up, device="cuda")
d_down = torch.empty_like(down, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def h2d():
    with torch.cuda.stream(s1):
        d_up.copy_(up, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        down.copy_(d_down, non_blocking=True)


def both():
    h2d(); d2h()


def h2d_chunks(n=4):
    step = up.numel() // n
    with torch.cuda.stream(s1):
        for k in range(n):
            d_up[k * step:(k + 1) * step].copy_(up[k * step:(k + 1) * step], non_blocking=True)


t_up, t_down, t_both, t_chunks = timed(h2d), timed(d2h), timed(both), timed(h2d_chunks)
print(f"H2D 104 MB: {t_up * 1e3:.2f} ms ({up.numel() / t_up / 1e9:.1f} GB/s); D2H 28 MB: {t_down * 1e3:.2f} ms "
      f"({down.numel() / t_down / 1e9:.1f} GB/s); both at once: {t_both * 1e3:.2f} ms; H2D in 4 pieces: {t_chunks * 1e3:.2f} ms")
