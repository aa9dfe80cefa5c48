"""CPU oracle for the dd3d gradient-matching hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package ``dd3d_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and there only as the checker /
the CPU baseline, never as the thing measured or shipped.

Parity status: the reference's own Python (``distillation_loss.py``,
``models/MEConv.py``, ``utils/Mnist2D.py``, ``utils/MinkowskiCollate.py``) is
pinned by golden vectors generated in the build container from the reference
files themselves (``tests/golden/make_golden.py``).  The MinkowskiEngine 0.5.4
semantics restated in ``oracle/minkowski_ref.py`` are **parity unpinned**: the
library is a third-party native dependency (``environment.yml:278``) that is
absent from /root/reference and not installable offline, and the reference has
no tests or fixtures at that boundary.

Two legs that share no code and are tested against each other (``tests/test_cpu_leg.py``): the torch restatement
(``minkowski_ref.py`` + ``ref_model.py``, what the golden fixtures pin) and a C++/OpenMP one (``cpu/dd3d_cpu.cpp`` behind
``cpu_leg.py``; built by ``__graft_entry__.build()`` / ``make -C oracle/cpu``; interface in ``cpu/dd3d_cpu.h``: one class of a step, and the
same restatement operator by operator on the array layouts of ``include/dd3d.h`` -- ``tests/test_cpu_leg_ops.py``).
"""
