"""CPU oracle for the reverse-SDE enhancement path of sp-uhh/sgmse.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``sgmse_amd``) never imports this package and fails loudly when the HIP
library is missing.

The oracle is a plain-PyTorch fp32 (CPU) restatement of the reference
algorithm, written functionally over a flat ``{state_dict_name: tensor}``
parameter dictionary.  Every function cites the reference ``file:line`` it
follows (paths relative to the upstream tree).

Parity pin: the reference ships no tests or golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against the *reference code
itself* executed on CPU in the build container by ``oracle/make_golden.py``;
the resulting input/output vectors are committed under ``tests/golden/`` and
``tests/test_oracle_golden.py`` re-checks the oracle against them everywhere
(the reference tree does not exist on the GPU box).
"""
