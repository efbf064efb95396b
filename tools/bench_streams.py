"""Experiment: the configs[1] step with the batch cut into S bundle groups, each group's launch chain on its own
HIP stream (S = 1 is bench.py's step).  The recurrent chain of a group is latency-bound (four dependent launches
per time step); independent chains fill each other's gaps.
    python tools/bench_streams.py [S ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from paccmann_polymer_amd.paccmann_encoder import Encoder
from paccmann_polymer_amd.optim import FlatAdam
from paccmann_polymer_amd.graph_constrained_loss import graph_loss_segmented


def main():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = Encoder(B.PARAMS, None, None, batch_size=1, batch_mode='Packed')
    model.gru_vae.train()
    vae = model.gru_vae
    opt = FlatAdam(vae.parameters(), lr=1e-4)
    rng = np.random.default_rng(1234)
    seqs, gs, sizes = B.synth_batch(rng, 128, 16)
    eps_all = torch.randn(len(seqs), B.PARAMS['latent_dim'], device=dev)
    total_valid = sum(len(s) for s in seqs) - len(seqs)
    for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
        groups = []
        per = 128 // S
        for g in range(S):
            b0, b1 = g * per, (g + 1) * per
            sq = seqs[b0 * 16:b1 * 16]
            enc, dec, tgt, g_flat = B.to_device_batch(sq, gs[b0:b1], dev)
            groups.append(dict(enc=enc, dec=dec, tgt=tgt, g=g_flat, sizes=sizes[b0:b1], eps=eps_all[b0 * 16:b1 * 16].contiguous(),
                               w=(sum(len(s) for s in sq) - len(sq)) / total_valid))
        model.update_batch_size(per * 16)
        streams = [torch.cuda.Stream() for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]

        def step():
            opt.zero_grad()
            main_s = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(main_s)
            mus = []
            for h, s in zip(groups, streams):                       # phase 1: encoders
                with torch.cuda.stream(s):
                    mu, lv = vae.encode(h['enc'])
                    mus.append((mu, lv, vae.reparameterize(mu, lv, h['eps'])))
            parts = []
            for h, s, (mu, lv, z) in zip(groups, streams, mus):     # phase 2: decoders + graph loss
                with torch.cuda.stream(s):
                    dl = vae.decode(z.unsqueeze(0), h['dec'], h['tgt'])
                    parts.append(dl * h['w'] + graph_loss_segmented(z, h['g'], h['sizes']))
            for s in streams:
                main_s.wait_stream(s)
            loss = parts[0]
            for p in parts[1:]:
                loss = loss + p
            loss.backward()
            opt.step()
            return loss.detach()

        import threading

        def worker(h, s, main_s, out, i):
            with torch.cuda.stream(s):
                mu, lv = vae.encode(h['enc'])
                z = vae.reparameterize(mu, lv, h['eps'])
                dl = vae.decode(z.unsqueeze(0), h['dec'], h['tgt'])
                loss = dl * h['w'] + graph_loss_segmented(z, h['g'], h['sizes'])
                loss.backward()
                out[i] = loss.detach()

        def step_threads():
            opt.zero_grad()
            main_s = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(main_s)
            out = [None] * S
            ths = [threading.Thread(target=worker, args=(h, s, main_s, out, i)) for i, (h, s) in enumerate(zip(groups, streams))]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            for s in streams:
                main_s.wait_stream(s)
            opt.step()
            return sum(out)

        if os.environ.get('PPK_STREAM_THREADS') == '1' and S > 1:
            step = step_threads
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f'streams {S}: host returns after {t_host * 1e3:.2f} ms, GPU done after {(time.perf_counter() - t0) * 1e3:.2f} ms', flush=True)
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            l = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f'streams {S}: {dt * 1e3:.2f} ms/step  {len(seqs) / dt:.0f} samples/s  loss {float(l):.4f}', flush=True)


main()
