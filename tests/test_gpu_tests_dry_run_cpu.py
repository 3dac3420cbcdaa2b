"""-m "not gpu": the BODIES of the model-level GPU tests executed with device = cpu on the CPU stand-ins of the ops
(tests/mock_backend.py).  Purpose: the python of those tests, their fixtures / goldens and the engine's host logic are
exercised in every CPU run, so that a GPU session is never spent on a typo or a stale threshold; the kernels
themselves are only tested when the same functions run with the real library (-m gpu).
"""
import pytest
import torch

import mock_backend

MODEL_TESTS = ["test_ptv3_tiny_forward_matches_reference_golden_and_oracle", "test_ptv3_two_scenes_forward_backward_vs_oracle",
               "test_ptv3_outdoor_depth12_four_channels", "test_ptv3_mix3d_duplicate_voxels",
               "test_ptv3_dense_rpe_branch_matches_reference_golden_and_oracle", "test_ptv3_enable_flash_false_uses_the_shrunk_patch",
               "test_ptv3_enc_mode_chain_matches_reference_golden_and_oracle", "test_ptv3m2_matches_reference_golden",
               "test_ptv3_pdnorm_ppt_configuration_matches_reference_golden"]
SPUNET_TESTS = ["test_spunet_tiny_matches_reference_golden_and_oracle", "test_spunet_base_channels_single_scene_and_duplicates",
                "test_spunet_enc_mode", "test_reference_style_model_file_runs_on_the_engine_through_compat"]


@pytest.mark.parametrize("name", MODEL_TESTS)
def test_gpu_model_test_bodies_on_cpu_standins(name):
    import test_gpu_model as T

    with mock_backend.cpu_ops():
        getattr(T, name)(torch.device("cpu"))


@pytest.mark.parametrize("name", SPUNET_TESTS)
def test_gpu_spunet_test_bodies_on_cpu_standins(name):
    import test_gpu_spunet as T

    with mock_backend.cpu_ops():
        getattr(T, name)(torch.device("cpu"))


FULLSIZE_TESTS = ["test_ptv3_base_one_full_scene_maps_and_logits", "test_ptv3_base_two_ragged_full_scenes_padding_borrow",
                  "test_ptv3_base_train_step_gradients_vs_oracle",
                  "test_spunet_base_one_full_scene_forward", "test_ptv3_outdoor_full_scene_forward",
                  "test_spunet_base_two_full_scenes_train_step_vs_oracle", "test_ptv3_base_b8_equals_the_sum_of_its_scenes",
                  "test_ptv3_outdoor_batch_equals_the_sum_of_its_scenes"]


@pytest.mark.parametrize("name", FULLSIZE_TESTS)
def test_gpu_fullsize_test_bodies_on_cpu_standins(name, monkeypatch):
    """the full-size / full-depth parity tests, bodies unchanged, on scenes shrunk 16x (PTC_FULLSIZE_SCALE) so that the
    base-depth models fit the CPU budget of this tier"""
    import importlib

    monkeypatch.setenv("PTC_FULLSIZE_SCALE", "0.0625")
    import test_gpu_fullsize as T

    T = importlib.reload(T)
    with mock_backend.cpu_ops():
        getattr(T, name)(torch.device("cpu"))


@pytest.mark.parametrize("name", ["test_ptv3m3_matches_reference_golden", "test_litept_matches_reference_golden"])
def test_gpu_m3_litept_test_bodies_on_cpu_standins(name):
    import test_gpu_m3_litept as T

    with mock_backend.cpu_ops():
        getattr(T, name)(torch.device("cpu"))


@pytest.mark.parametrize("family", ["m2", "m3", "litept"])
def test_f2_models_at_shipped_widths_body_on_cpu_standins(family, monkeypatch):
    """the 48 .. 576-channel configurations of the reference's m2 / m3 / LitePT configs construct and step (host logic, padding rules)"""
    import test_gpu_m3_litept as T

    with mock_backend.cpu_ops():
        T.test_f2_models_at_shipped_widths_run_on_the_engine_only(torch.device("cpu"), family, monkeypatch)


@pytest.mark.parametrize("name", ["test_sync_bn_conversion_ptv3_keeps_every_activation",
                                  "test_sync_bn_conversion_spunet_degrades_to_the_three_pass_block",
                                  "test_sync_bn_conversion_litept_matches_the_reference_golden",
                                  "test_wrapped_or_hooked_modules_are_never_fused",
                                  "test_spunet_block_tail_is_not_fused_when_bn2_is_hooked"])
def test_gpu_sync_bn_test_bodies_on_cpu_standins(name):
    """nn.SyncBatchNorm.convert_sync_batchnorm (reference trainer, sync_bn=True) on the engine's models"""
    import test_gpu_sync_bn as T

    with mock_backend.cpu_ops():
        getattr(T, name)(torch.device("cpu"))
