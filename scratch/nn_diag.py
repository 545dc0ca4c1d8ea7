import sys, time, torch
sys.path[:0] = [".", "anomaly-clustering_amd/models", "tests"]
import acnative
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "randn"
if which == "randn":
    Z = torch.randn(83, 784, 4096, generator=g).to(dev); B = torch.randn(83, 784, 4096, generator=g).to(dev)
else:   # real pipeline embeddings (random-init ViT on synthetic images), test vs train
    import os
    os.environ.setdefault("AC_BACKBONE_SEED", "0")
    import patchcore.backbones, patchcore.common as common, patchcore.sampler
    from patchcore.patchcore import AnomalyClusteringCore
    from patchcore.datasets import synthetic
    bb = patchcore.backbones.load("dino_vitbase8")
    core = AnomalyClusteringCore(dev)
    core.load(backbone=bb, layers_to_extract_from=["blocks.10", "blocks.11"], device=dev, input_shape=(3, 224, 224),
              pretrain_embed_dimension=2048, target_embed_dimension=4096, patchsize=3,
              featuresampler=patchcore.sampler.IdentitySampler(), nn_method=common.FaissNN(False, 4))
    Z = core.embed_device(synthetic.make_images("bottle", synthetic.category_labels("bottle")).to(dev), 128)
    B = core.embed_device(synthetic.make_images("bottle", ["good"] * 83, split=synthetic.DatasetSplit.TRAIN).to(dev), 128)
q, k = acnative.pack_images(Z), acnative.pack_images(B)
q16, k16 = acnative.h16_from_packed(q), acnative.h16_from_packed(k)
slack = acnative.nn_slack(acnative.max_sq([k], dev), acnative.max_sq([q], dev))
print("slack", slack.tolist(), "max_sq", acnative.max_sq([k], dev).item())
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for cap in (32, 256):
    st = acnative.NNState(q.n * q.P, dev, cap)
    lib = acnative.lib()
    for rep in range(2):
        st = acnative.NNState(q.n * q.P, dev, cap)
        e0 = ev()
        acnative._check(lib.ac_nn_candidates(k16.data_ptr(), k.sq.data_ptr(), k.n * k.P, q16.data_ptr(), q.sq.data_ptr(), q.n * q.P, q.D,
                        slack.data_ptr(), st.approx_min.data_ptr(), st.cand.data_ptr(), st.count.data_ptr(), st.cap, 0, acnative._stream()))
        e1 = ev()
        acnative._check(lib.ac_nn_refine(k.hl.data_ptr(), k.n * k.P, q.hl.data_ptr(), q.n * q.P, q.D, st.cand.data_ptr(), st.count.data_ptr(),
                        st.cap, 0, st.best.data_ptr(), st.overflow.data_ptr(), acnative._stream()))
        e2 = ev(); torch.cuda.synchronize()
    c = st.count.float()
    print("cap", cap, "pass1 %.1f ms  refine %.1f ms" % (e0.elapsed_time(e1), e1.elapsed_time(e2)),
          "cand mean %.1f  p50 %.0f  p99 %.0f  max %.0f  overflow %d" % (c.mean().item(), c.median().item(), c.quantile(0.99).item(), c.max().item(), st.overflow.item()))
