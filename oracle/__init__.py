"""CPU oracle for the Sat-NeRF volumetric-rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``satnerf_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker / the reported CPU
baseline -- never as the thing that is measured or shipped.
"""
