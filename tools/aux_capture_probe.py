"""Does a fork/join onto a second stream survive HIP-graph capture on this ROCm?  Each case in its own child process (a failing
capture_end segfaults); the parent never touches the GPU.  Usage: python tools/aux_capture_probe.py > gpurun_out/r05/aux_probe.txt"""
import os, subprocess, sys

CASES = ["torch_fwd", "torch_bwd", "torch_bwd_leaf_on_aux", "step:0", "step:roi", "step:gnn", "step:1"]


def child(case):
    import faulthandler; faulthandler.enable()
    import torch
    dev = torch.device("cuda:0")
    if case.startswith("torch"):
        side, aux = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        w1 = torch.randn(256, 256, device=dev, requires_grad=True)
        w2 = torch.randn(256, 256, device=dev, requires_grad=True)
        x = torch.randn(64, 256, device=dev)

        def body():
            h = x @ w1
            if case != "torch_fwd":
                h = torch.tanh(h)
            aux.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(aux):
                a = torch.relu(h @ w2) if case == "torch_bwd_leaf_on_aux" else torch.relu(h * 2.0)
            b = h @ w1
            torch.cuda.current_stream().wait_stream(aux)
            out = (a + b).sum()
            if case != "torch_fwd":
                w1.grad = w2.grad = None
                out.backward()
            return out
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = body()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print(case, "ok", float(out))
        return
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import copy
    from tests.test_gpu_optim import _tiny_cfg, GeneratorFullModel, TrainStep, batch_to, make_batch, make_step_rng
    cfg = _tiny_cfg(); tp = cfg["train_params"]
    torch.manual_seed(0)
    model = GeneratorFullModel(train_params=copy.deepcopy(tp), model_params=copy.deepcopy(cfg["model_params"]),
                               dataset="cityscapes").to(dev).train()
    step = TrainStep(model, run_optimizers=True, distributed=False)
    batch = batch_to(make_batch(1, 128, 256, 2, seed=51), dev)
    rng = make_step_rng(batch, z_dim=16, latent_dim=32, seed=0)
    batch["rng"] = {k: v.to(dev) for k, v in rng.items()}
    step.capture(batch)
    tot = []
    for _ in range(3):
        _, lg, _ = step(batch)
        tot.append(float(lg["total_gen"].detach()))
    torch.cuda.synchronize()
    print(case, "ok", tot)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for c in CASES:
            env = dict(os.environ)
            if c.startswith("step:"):
                env["C2M_AUX_STREAM"] = c.split(":")[1]
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], env=env, capture_output=True, text=True, timeout=600)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            err = [l for l in r.stderr.splitlines() if "Fatal" in l or "Error" in l or "error" in l][:3]
            print(f"{c:28s} rc={r.returncode:4d} {tail} {err}", flush=True)
