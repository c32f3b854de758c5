"""CPU oracle for the SeMIGCN graph-convolution hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``semigcn_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the reported baseline.

Contents
--------
``pyg_restatement``  plain-torch (CPU, ATen) restatement of the torch-geometric
                     2.2.0 / torch-scatter 2.1.0 operators the reference calls
                     (``ChebConv``, ``Sequential``, ``Data``) -- third-party code
                     that is NOT vendored under /root/reference.
``models``           restatement of the reference's own model composition
                     (``SingleScaleGCN``, ``MeshPool``/``MeshUnpool``,
                     ``DownConv``/``UpConv``, ``MGCN.forward``) on top of it.
``dense``            fp64 dense ``-D^-1/2 A D^-1/2`` evaluation, the accuracy arbiter.
``ref_shim``         container-only loader that imports the *reference's own*
                     ``util/*.py`` from /root/reference (never copied) on top of
                     ``pyg_restatement``; used by ``make_golden.py`` to freeze
                     fixtures into ``tests/golden``.

Pinning status: the reference has no tests, golden vectors or fixtures for this
path and torch-geometric is absent, so the [3P] operator
semantics are **parity unpinned by the reference's own tests**.  They are pinned
instead by (i) the reference's own ``Mesh.build_adj_mat`` matrices
(util/mesh.py:276-285) evaluated densely in fp64, (ii) the reference's own
module code (util/networks.py, util/meshnet.py) run through ``ref_shim`` to
produce the committed golden vectors.
"""
