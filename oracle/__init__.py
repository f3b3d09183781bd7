"""CPU oracle for the CodeFormer aligned-face path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``codeformer_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do, and only as the checker / the timed CPU baseline.
"""
