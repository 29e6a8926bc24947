"""The helper that starts rank processes for the multi-process GPU tests (tests/rank_spawner.py, conftest `spawn_ranks`),
exercised on CPU: rank environment, captured output, failure and timeout handling."""


def test_ranks_get_their_environment(spawn_ranks):
    res = spawn_ranks(2, ["-c", "import os; print('rank', os.environ['RANK'], 'of', os.environ['WORLD_SIZE'], os.environ['MASTER_ADDR'], "
                                "os.path.basename(os.environ['LIGHTGRAD_RCCL_ID_FILE']), os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])"], timeout=60)
    assert res["rc"] == 0 and res["codes"] == [0, 0]
    assert res["outputs"][0].strip() == "rank 0 of 2 127.0.0.1 rccl.id 0"
    assert res["outputs"][1].strip() == "rank 1 of 2 127.0.0.1 rccl.id 0"


def test_a_failing_rank_fails_the_job_and_the_other_is_stopped(spawn_ranks):
    res = spawn_ranks(2, ["-c", "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(60)"], timeout=60)
    assert res["rc"] == 7
    assert res["codes"][1] == 7 and res["codes"][0] not in (0, None)


def test_timeout(spawn_ranks):
    res = spawn_ranks(1, ["-c", "import time; time.sleep(60)"], timeout=1.0)
    assert res["rc"] == 124


def test_per_rank_environment_and_wait_for_all(spawn_ranks):
    res = spawn_ranks(2, ["-c", "import os, sys; print(os.environ['COLOUR']); sys.exit(3 if os.environ['RANK'] == '0' else 0)"],
                      env={"COLOUR": "red"}, rank_env={"1": {"COLOUR": "blue"}}, timeout=60, wait_for_all=True)
    assert res["rc"] == 3 and res["codes"] == [3, 0]
    assert [o.strip() for o in res["outputs"]] == ["red", "blue"]
