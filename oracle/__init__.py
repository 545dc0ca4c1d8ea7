"""CPU oracle for the anomaly-clustering hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU fp32 / numpy fp64) of the
reference's arithmetic for the path SURVEY.md section 8 scopes:

    ViT-B/8 forward -> patch aggregation -> alpha weights -> weighted average
    -> pairwise distance -> Ward clustering -> NMI / ARI / F1

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker (or as the
CPU baseline that is timed *beside* the GPU path).  Nothing under
``anomaly-clustering_amd/`` imports it; the product path fails loudly when
the HIP library is missing instead of falling back to this code.

Pinning status (see DESIGN.md "Oracle"):
  * cluster/metrics stage  - pinned by the reference's own committed
    artefacts (26 known-answer CSV rows; fixtures under tests/golden/kat_*).
  * ViT / aggregation / alpha / weighted-average stages - the reference holds
    no golden vectors for these; they are pinned by outputs of the reference
    itself, imported read-only in the build container by
    ``tests/golden/make_golden.py`` (which is committed next to the vectors
    it wrote).

  * image transform (Resize / CenterCrop / ToTensor / Normalize) - torchvision is absent; pinned one level
    down, against Pillow itself (the arithmetic torchvision runs for PIL inputs) and a committed fixture.

Every function cites the reference file:line it restates; paths are relative
to ``/root/reference/Anomaly-Clustering/``.
"""

from . import vit, aggregate, alpha, cluster, preprocess  # noqa: F401
