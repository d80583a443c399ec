"""The interactive camera of the reference (Camera.h:16-139) mirrored by rt_camera_rig_*:
state transitions, clamps, sensitivity and the LoopNum reset on every change."""
import math

import numpy as np
import pytest

import raytracing_amd as rt


def _vec(a):
    return np.array(list(a), dtype=np.float32)


def test_initial_state_matches_camera_constructor():
    rig = rt.CameraRig(1200, 800)
    s = rig.s
    assert (s.Yaw, s.Pitch, s.fov, s.cameraSpeed, s.firstMouse, s.LoopNum) == (-90.0, 0.0, 45.0, 1.0, 1, 0)
    assert list(s.cameraPos) == [0.0, 2.0, 5.0] and list(s.worldUp) == [0.0, 1.0, 0.0]
    assert s.ScreenRatio == np.float32(1200) / np.float32(800)
    assert s.halfH == pytest.approx(1.0, abs=1e-6) and s.halfW == pytest.approx(1.5, abs=1e-6)
    # the same frame as rt_camera_from_angles with the constructor's values
    cam = rt.camera_from_angles((0, 2, 5), -90.0, 0.0, 45.0, 1200, 800)
    u = rig.uniforms()
    for f in ("camPos", "front", "right", "up", "leftbottom", "halfH", "halfW"):
        assert np.array_equal(u[f], cam[f]), f
    assert np.allclose(u["front"][0], [0, 0, -1], atol=1e-6)
    assert np.allclose(u["leftbottom"][0], [-1.5, -1.0, -1.0], atol=1e-6)


def test_fov_is_clamped_and_resets_the_loop():
    rig = rt.CameraRig(800, 800)
    rig.loop_increase(); rig.loop_increase()
    assert rig.s.LoopNum == 2
    rig.update_fov(10.0)                       # scroll up narrows the view: fov -= offset
    assert rig.s.fov == 35.0 and rig.s.LoopNum == 0
    assert rig.s.halfH == pytest.approx(math.tan(math.radians(35.0)), rel=1e-6)
    assert rig.s.halfW == pytest.approx(rig.s.halfH, rel=1e-7)
    rig.update_fov(100.0)
    assert rig.s.fov == 1.0
    rig.update_fov(-100.0)
    assert rig.s.fov == 45.0
    lb = _vec(rig.s.cameraFront) - rig.s.halfW * _vec(rig.s.cameraRight) - rig.s.halfH * _vec(rig.s.cameraUp)
    assert np.allclose(_vec(rig.s.LeftBottomCorner), lb, atol=1e-6)


def test_mouse_look_sensitivity_first_event_and_pitch_clamp():
    rig = rt.CameraRig(1200, 800)
    rig.loop_increase()
    rig.update_front(400.0, 300.0)             # first event only records the position
    assert (rig.s.Yaw, rig.s.Pitch, rig.s.firstMouse, rig.s.LoopNum) == (-90.0, 0.0, 0, 0)
    rig.update_front(500.0, 200.0)             # 100 px right, 100 px up: 3 degrees each
    assert rig.s.Yaw == pytest.approx(-87.0, abs=1e-4) and rig.s.Pitch == pytest.approx(3.0, abs=1e-4)
    f = _vec(rig.s.cameraFront)
    want = [math.cos(math.radians(-87)) * math.cos(math.radians(3)), math.sin(math.radians(3)),
            math.sin(math.radians(-87)) * math.cos(math.radians(3))]
    assert np.allclose(f, want, atol=1e-6)
    assert np.dot(f, _vec(rig.s.cameraRight)) == pytest.approx(0.0, abs=1e-6)
    assert np.allclose(np.cross(_vec(rig.s.cameraRight), f), _vec(rig.s.cameraUp), atol=1e-6)
    rig.update_front(500.0, -100000.0)
    assert rig.s.Pitch == 89.0
    rig.update_front(500.0, 100000.0)
    assert rig.s.Pitch == -89.0


def test_keyboard_moves_along_front_and_right():
    rig = rt.CameraRig(1200, 800)
    p0 = _vec(rig.s.cameraPos)
    rig.loop_increase()
    rig.process_keyboard(rt.CameraRig.FORWARD, 0.5)
    assert np.allclose(_vec(rig.s.cameraPos), p0 + 0.5 * _vec(rig.s.cameraFront), atol=1e-6) and rig.s.LoopNum == 0
    rig.process_keyboard(rt.CameraRig.BACKWARD, 0.5)
    assert np.allclose(_vec(rig.s.cameraPos), p0, atol=1e-6)
    rig.process_keyboard(rt.CameraRig.RIGHT, 2.0)
    assert np.allclose(_vec(rig.s.cameraPos), p0 + 2.0 * _vec(rig.s.cameraRight), atol=1e-6)
    rig.process_keyboard(rt.CameraRig.LEFT, 2.0)
    assert np.allclose(_vec(rig.s.cameraPos), p0, atol=1e-6)
    with pytest.raises(rt.RtError):
        rig.process_keyboard(7, 1.0)


def test_resize_changes_only_the_horizontal_extent():
    rig = rt.CameraRig(1200, 800)
    h = rig.s.halfH
    rig.loop_increase()
    rig.update_screen_ratio(800, 800)
    assert rig.s.halfH == h and rig.s.halfW == pytest.approx(h, rel=1e-7) and rig.s.LoopNum == 0
    with pytest.raises(rt.RtError):
        rig.update_screen_ratio(0, 10)
