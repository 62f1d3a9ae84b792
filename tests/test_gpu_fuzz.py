"""A fixed range of the seeds of tests/fuzz_parity.py with the GPU suite: random scenes, cameras, parameters and ray batches
under random scheduling settings of the traversal, HIP path against the CPU restatement, bit for bit.  Needs an MI355X: run
with -m gpu."""
import pytest

pytestmark = pytest.mark.gpu


def test_random_scenes_cameras_and_schedules_equal_the_restatement():
    import fuzz_parity
    failed = fuzz_parity.run(0, 60)
    assert not failed, "\n".join(failed)


def test_random_large_frames_equal_the_binary_tree_traversal_and_the_restatement():
    import fuzz_parity
    failed = fuzz_parity.run(0, 8, scene=fuzz_parity.one_scene_big)
    assert not failed, "\n".join(failed)


def test_random_ways_of_cutting_a_call_up_give_one_image():
    import fuzz_parity
    failed = fuzz_parity.run(0, 10, scene=fuzz_parity.one_scene_batching)
    assert not failed, "\n".join(failed)
