"""GPU parity for the device-resident PageState kernels: the reference's page-manager scenarios
(tests/inference/page_manager_test.py) and a random operation sequence, each compared field by field with the oracle."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

FIELDS = ("page_status", "page_map", "num_pages_used", "sequence_lengths", "active_page", "has_active_page", "active_page_position")
NUM_PAGES, TPP, MAX_TARGET, GROUPS = 128, 8, 256, 4


@pytest.fixture(scope="module")
def pair():
  from oracle import oracle as O
  cfg = H.make_config(per_device_batch_size=float(GROUPS), pagedattn_num_pages=NUM_PAGES, pagedattn_tokens_per_page=TPP,
                      max_prefill_predict_length=128, max_target_length=MAX_TARGET)
  maxengine = H.importlib.import_module("maxtext-indextts2_amd.maxengine")
  eng = maxengine.MaxEngine(cfg)  # no weights needed for page-state operations
  opm = O.PageManager(NUM_PAGES, TPP, MAX_TARGET, GROUPS, cfg.pagedattn_max_pages_per_group)
  return eng, opm, O


def same(eng, ost):
  st = eng.page_manager.device_state().to_numpy()
  for f in FIELDS:
    assert np.array_equal(st[f], getattr(ost, f)), (f, st[f], getattr(ost, f))


def test_reference_scenarios(pair):
  eng, opm, O = pair
  pm = eng.page_manager
  st = opm.get_initial_page_state()
  pm.get_initial_page_state()
  same(eng, st)
  # decode with nothing active changes nothing (:233-236)
  pm.update_decode_pages(None); st = opm.update_decode_pages(st); same(eng, st)
  # reserve / edge cases / boundary crossing (:93-139, :162-183, :469-500)
  for g, n in [(0, 12), (1, TPP * 2), (2, 5), (3, TPP)]:
    pm.update_prefill_pages(None, g, n); st = opm.update_prefill_pages(st, g, n); same(eng, st)
  for _ in range(3):
    pm.update_decode_pages(None); st = opm.update_decode_pages(st); same(eng, st)
  # release one group, others untouched (:405-421)
  pm.release_pages(None, 1); st = opm.release_pages(st, 1); same(eng, st)
  # re-reserve replaces the group's pages (release + reserve, page_manager.py:314-328)
  pm.update_prefill_pages(None, 0, 40); st = opm.update_prefill_pages(st, 0, 40); same(eng, st)
  # max length (:352-368)
  pm.update_prefill_pages(None, 2, MAX_TARGET); st = opm.update_prefill_pages(st, 2, MAX_TARGET); same(eng, st)
  # decode at max_pages_per_group: no capacity, length still advances
  pm.update_decode_pages(None); st = opm.update_decode_pages(st); same(eng, st)


def test_no_space(pair):
  eng, opm, O = pair
  pm = eng.page_manager
  st = opm.get_initial_page_state()
  ps = pm.get_initial_page_state()
  st.page_status[:] = 1
  ps.replace(page_status=np.ones((NUM_PAGES,), np.int32))
  pm.update_prefill_pages(None, 0, 12); st = opm.update_prefill_pages(st, 0, 12); same(eng, st)  # (:141-160)
  # partially free pool: exactly enough, then one short
  st.page_status[10:12] = 0
  ps.replace(page_status=st.page_status)
  pm.update_prefill_pages(None, 1, 16); st = opm.update_prefill_pages(st, 1, 16); same(eng, st)
  pm.update_prefill_pages(None, 2, 9); st = opm.update_prefill_pages(st, 2, 9); same(eng, st)
  # decode step that needs a page while none is free: silent (page_manager.py:373-376)
  pm.update_decode_pages(None); st = opm.update_decode_pages(st); same(eng, st)


def test_invalid_arguments(pair):
  eng, _, _ = pair
  pm = eng.page_manager
  for bad in (-1, GROUPS):
    with pytest.raises(ValueError):
      pm.update_prefill_pages(None, bad, 10)
    with pytest.raises(ValueError):
      pm.release_pages(None, bad)
  for bad in (-1, 0, MAX_TARGET + 1):
    with pytest.raises(ValueError):
      pm.update_prefill_pages(None, 0, bad)


def test_random_operation_sequence(pair):
  eng, opm, O = pair
  pm = eng.page_manager
  rng = np.random.default_rng(42)
  st = opm.get_initial_page_state()
  pm.get_initial_page_state()
  for it in range(300):
    op = rng.integers(0, 10)
    g = int(rng.integers(0, GROUPS))
    if op < 2:
      n = int(rng.integers(1, MAX_TARGET + 1))
      pm.update_prefill_pages(None, g, n); st = opm.update_prefill_pages(st, g, n)
    elif op < 3:
      pm.release_pages(None, g); st = opm.release_pages(st, g)
    elif op < 4 and st.has_active_page[g] and st.sequence_lengths[g] < MAX_TARGET - 20:
      n = int(st.sequence_lengths[g] + rng.integers(1, 20))
      pm.extend_prefill_pages(None, g, n); st = opm.extend_prefill_pages(st, g, n)
    else:
      pm.update_decode_pages(None); st = opm.update_decode_pages(st)
    if it % 10 == 0:
      same(eng, st)
  same(eng, st)
