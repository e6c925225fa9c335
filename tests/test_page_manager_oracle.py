"""The reference's own page-manager test cases (MaxText/tests/inference/page_manager_test.py:29-541), re-stated against
the oracle's PageManager.  These exact-integer cases are what PINS the oracle's page-manager restatement; the same
scenarios run against the HIP page-state kernels in test_gpu_pages.py."""
import numpy as np
import pytest

from oracle import oracle as O

NUM_PAGES, TPP, MAX_PREFILL, MAX_TARGET = 128, 8, 128, 256  # page_manager_test.py:32-36
MPG = (MAX_TARGET + TPP - 1) // TPP
GROUPS = 4


@pytest.fixture
def pm():
  return O.PageManager(NUM_PAGES, TPP, MAX_TARGET, GROUPS, MPG)


def eq(a: O.PageState, b: O.PageState):
  return all(np.array_equal(getattr(a, f), getattr(b, f)) for f in a.__dataclass_fields__)


def test_initialization(pm):  # :80-91
  s = pm.get_initial_page_state()
  assert s.page_status.shape == (NUM_PAGES,) and s.page_map.shape == (GROUPS, MPG)
  assert np.all(s.page_status[1:] == 0) and s.page_status[0] == 1
  assert np.all(s.num_pages_used == 0) and np.all(s.sequence_lengths == 0) and not np.any(s.has_active_page)


def test_reserve_prefill_group(pm):  # :93-139
  s = pm.update_prefill_pages(pm.get_initial_page_state(), 0, 12)
  need = 2
  assert s.sequence_lengths[0] == 12 and s.num_pages_used[0] == need and s.has_active_page[0]
  used = s.page_map[0, :need]
  assert len(np.unique(used)) == need and np.all(s.page_status[used] == 1)
  assert s.active_page_position[0] == 12 % TPP
  assert list(used) == [1, 2]  # first fit from page 1 (page 0 reserved, page_manager.py:115-117,145-146)


def test_reserve_prefill_no_space(pm):  # :141-160
  s = pm.get_initial_page_state()
  s.page_status[:] = 1
  u = pm.update_prefill_pages(s, 0, 12)
  assert np.all(u.page_status == 1) and u.sequence_lengths[0] == 0 and u.num_pages_used[0] == 0 and not u.has_active_page[0]


def test_reserve_prefill_edge_cases(pm):  # :162-183
  init = pm.get_initial_page_state()
  a = pm.update_prefill_pages(init, 1, TPP * 2)
  assert (a.sequence_lengths[1], a.num_pages_used[1], a.active_page_position[1]) == (16, 2, 0) and a.has_active_page[1]
  b = pm.update_prefill_pages(init, 2, 5)
  assert (b.sequence_lengths[2], b.num_pages_used[2], b.active_page_position[2]) == (5, 1, 5) and b.has_active_page[2]


def test_release_pages(pm):  # :185-228
  a = pm.update_prefill_pages(pm.get_initial_page_state(), 1, 10)
  pages = a.page_map[1, :a.num_pages_used[1]].copy()
  before = int(a.page_status.sum())
  r = pm.release_pages(a, 1)
  assert r.sequence_lengths[1] == 0 and r.num_pages_used[1] == 0 and not r.has_active_page[1]
  assert int(r.page_status.sum()) == before - len(pages) and np.all(r.page_status[pages] == 0)


def test_update_decode_pages(pm):  # :230-286
  init = pm.get_initial_page_state()
  assert eq(init, pm.update_decode_pages(init))
  p = pm.update_prefill_pages(init, 0, TPP)
  assert (p.num_pages_used[0], p.sequence_lengths[0], p.active_page_position[0]) == (1, TPP, 0)
  d = pm.update_decode_pages(p)
  assert d.sequence_lengths[0] == TPP + 1 and d.num_pages_used[0] == 2
  assert d.page_map[0, 1] != p.page_map[0, 0] and d.active_page[0] == d.page_map[0, 1] and d.active_page_position[0] == 0
  q = pm.update_prefill_pages(init, 1, 5)
  first = q.active_page[1]
  e = pm.update_decode_pages(q)
  assert (e.sequence_lengths[1], e.num_pages_used[1], e.active_page[1], e.active_page_position[1]) == (6, 1, first, 5)


def test_state_consistency(pm):  # :288-335
  init = pm.get_initial_page_state()
  assert int(init.page_status.sum()) == 1
  a = pm.update_prefill_pages(init, 0, 12)
  assert int(a.page_status.sum()) == int(a.num_pages_used.sum()) + 1
  d = pm.update_decode_pages(a)
  assert int(d.page_status.sum()) == int(d.num_pages_used.sum()) + 1
  r = pm.release_pages(d, 0)
  assert int(r.page_status.sum()) == int(r.num_pages_used.sum()) + 1 and r.num_pages_used[0] == 0


def test_page_group_id_boundaries(pm):  # :337-368
  init = pm.get_initial_page_state()
  a = pm.update_prefill_pages(init, GROUPS - 1, 1)
  assert a.sequence_lengths[GROUPS - 1] == 1 and a.num_pages_used[GROUPS - 1] == 1 and a.has_active_page[GROUPS - 1]
  b = pm.update_prefill_pages(init, 0, MAX_TARGET)
  assert b.sequence_lengths[0] == MAX_TARGET and b.num_pages_used[0] == MPG and b.has_active_page[0]


def test_multiple_group_interactions(pm):  # :370-421
  s = pm.get_initial_page_state()
  for g in range(3):
    n = 10 + g * TPP
    s = pm.update_prefill_pages(s, g, n)
    assert s.sequence_lengths[g] == n and s.num_pages_used[g] == (n + TPP - 1) // TPP and s.has_active_page[g]
    assert int(s.page_status.sum()) == int(s.num_pages_used.sum()) + 1
  d = pm.update_decode_pages(s)
  for g in range(3):
    n = 10 + g * TPP
    assert d.sequence_lengths[g] == n + 1 and d.num_pages_used[g] == (n + 1 + TPP - 1) // TPP
  r = pm.release_pages(d, 1)
  assert r.num_pages_used[1] == 0 and not r.has_active_page[1]
  for g in (0, 2):
    assert r.sequence_lengths[g] == d.sequence_lengths[g] and r.num_pages_used[g] == d.num_pages_used[g] and r.has_active_page[g]


def test_invalid_page_group_id(pm):  # :423-451
  init = pm.get_initial_page_state()
  for bad in (-1, GROUPS):
    with pytest.raises(ValueError):
      pm.update_prefill_pages(init, bad, 10)
    with pytest.raises(ValueError):
      pm.release_pages(init, bad)


def test_invalid_length(pm):  # :453-467
  init = pm.get_initial_page_state()
  for bad in (-1, 0, MAX_TARGET + 1):
    with pytest.raises(ValueError):
      pm.update_prefill_pages(init, 0, bad)


def test_page_boundary_crossing(pm):  # :469-500
  s = pm.update_prefill_pages(pm.get_initial_page_state(), 0, TPP)
  u = pm.update_decode_pages(s)
  assert u.num_pages_used[0] == 2 and u.sequence_lengths[0] == TPP + 1 and u.active_page_position[0] == 0
  assert u.active_page[0] != s.active_page[0]


def test_repeated_allocation_deallocation(pm):  # :502-541
  s = pm.get_initial_page_state()
  for i in range(3):
    for g in range(3):
      n = (g + 1) * 3 + i
      s = pm.update_prefill_pages(s, g, n)
      assert s.sequence_lengths[g] == n and s.has_active_page[g]
    for _ in range(2):
      s = pm.update_decode_pages(s)
    for g in range(3):
      s = pm.release_pages(s, g)
      assert s.sequence_lengths[g] == 0 and s.num_pages_used[g] == 0 and not s.has_active_page[g]
    assert int(s.num_pages_used.sum()) == 0 and int(s.page_status.sum()) == 1
  assert np.all(s.page_status[1:] == 0)


def test_init_validation():  # page_manager.py:450-473
  with pytest.raises(ValueError):
    O.PageManager(NUM_PAGES, TPP, MAX_TARGET, GROUPS, 0)
  with pytest.raises(ValueError):
    O.PageManager(NUM_PAGES, TPP, MAX_TARGET, GROUPS, MPG - 1)
  with pytest.raises(ValueError):
    O.PageManager(1, TPP, MAX_TARGET, GROUPS, MPG)
  with pytest.raises(ValueError):
    O.PageManager(NUM_PAGES, TPP, MAX_TARGET, 0, MPG)


def test_decode_without_free_page_is_silent(pm):  # page_manager.py:373-376: allocation fails, length still advances
  s = pm.update_prefill_pages(pm.get_initial_page_state(), 0, TPP)
  s.page_status[:] = 1
  u = pm.update_decode_pages(s)
  assert u.sequence_lengths[0] == TPP + 1 and u.num_pages_used[0] == 1 and u.active_page[0] == s.active_page[0]


def test_extend_matches_single_reserve(pm):
  """Build extension: reserve(n1) + extend(n1+n2) gives the page set a single reserve(n1+n2) gives on an empty pool."""
  a = pm.update_prefill_pages(pm.get_initial_page_state(), 0, 20)
  a = pm.extend_prefill_pages(a, 0, 45)
  b = pm.update_prefill_pages(pm.get_initial_page_state(), 0, 45)
  assert eq(a, b)
