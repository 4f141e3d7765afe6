"""episodes leave the host mirror in the reference's replay-buffer format.

The bytes written by muzero_amd/host/game_buffer_proto.cpp are decoded here by an independent protobuf
implementation (google.protobuf) whose message classes are built from the reference's schema
(platform/src/main/proto/.../GameBuffer.proto:11-66, restated below field by field), compared value by value with a
JSON dump of the same episodes, and serialised again: protobuf's canonical field order makes that byte-identical.
The container is checked with zipfile ("<network>-protobuf.zip" / "<network>.dat", GameBufferIO.java:129-170).
The reference holds no protobuf fixture of its own (its only buffer file, games/pegsolitair/src/test/resources/
games/buffer1000.zip, is the older JSON format) => parity with protobuf-java's output is pinned by the wire-format
specification, not by a golden file.
"""
import json
import os
import subprocess
import zipfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "muzero_amd", "host", "bin", "gamebuffer_tool")

pb = pytest.importorskip("google.protobuf")


def _messages():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="GameBuffer.proto", syntax="proto3",
                                            package="ai.enpasos.muzero.platform.agent.memory.protobuf")

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, rep, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=F.LABEL_REPEATED if rep else F.LABEL_OPTIONAL)
            if tname:
                f.type_name = "." + fd.package + "." + tname

    msg("PolicyProtos", [("policy", 1, F.TYPE_FLOAT, True, None)])
    msg("ObservationProtos", [("observationPartA", 1, F.TYPE_BYTES, False, None), ("observationPartB", 2, F.TYPE_BYTES, False, None),
                              ("observationPartSize", 3, F.TYPE_INT32, False, None), ("twoPlayer", 4, F.TYPE_BOOL, False, None)])
    msg("LegalActionProtos", [("legal_action", 1, F.TYPE_BOOL, True, None)])
    msg("GameProto", [
        ("actions", 1, F.TYPE_INT32, True, None), ("rewards", 2, F.TYPE_FLOAT, True, None),
        ("policy_targets", 3, F.TYPE_MESSAGE, True, "PolicyProtos"), ("playout_policy", 25, F.TYPE_MESSAGE, True, "PolicyProtos"),
        ("observations", 27, F.TYPE_MESSAGE, True, "ObservationProtos"), ("root_value_targets", 4, F.TYPE_FLOAT, True, None),
        ("v_mix", 31, F.TYPE_FLOAT, True, None), ("root_values_from_initial_inference", 5, F.TYPE_FLOAT, True, None),
        ("root_entropy_value_targets", 29, F.TYPE_FLOAT, True, None),
        ("root_entropy_values_from_initial_inference", 30, F.TYPE_FLOAT, True, None),
        ("last_value_error", 6, F.TYPE_FLOAT, False, None), ("entropies", 7, F.TYPE_FLOAT, True, None),
        ("count", 8, F.TYPE_INT64, False, None), ("tdSteps", 9, F.TYPE_INT32, False, None),
        ("trainingEpoch", 26, F.TYPE_INT32, False, None), ("surprises", 10, F.TYPE_FLOAT, True, None),
        ("surprised", 11, F.TYPE_BOOL, False, None), ("networkName", 15, F.TYPE_STRING, False, None),
        ("pRandomActionRawSum", 16, F.TYPE_FLOAT, False, None), ("pRandomActionRawCount", 17, F.TYPE_INT32, False, None),
        ("legal_actions", 18, F.TYPE_MESSAGE, True, "LegalActionProtos"), ("nextSurpriseCheck", 20, F.TYPE_INT64, False, None),
        ("hybrid", 21, F.TYPE_BOOL, False, None), ("tHybrid", 22, F.TYPE_INT64, False, None),
        ("maxEntropies", 24, F.TYPE_FLOAT, True, None)])
    msg("GameBufferProto", [("version", 10, F.TYPE_INT32, False, None), ("game_class_name", 11, F.TYPE_STRING, False, None),
                            ("counter", 12, F.TYPE_INT32, False, None), ("game_protos", 1, F.TYPE_MESSAGE, True, "GameProto")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName(fd.package + ".GameBufferProto"))


def _bits(b, n):
    """java.util.BitSet.valueOf(bytes) -> first n bits"""
    return [(b[i // 8] >> (i % 8)) & 1 if i // 8 < len(b) else 0 for i in range(n)]


def test_buffer_bytes_decode_with_an_independent_protobuf_and_reserialise_identically(tmp_path):
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "muzero_amd", "host"), "-s"])
    out = subprocess.run([BIN, str(tmp_path), "40", "5"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr   # includes the tool's own zip round trip
    raw = (tmp_path / "buffer.dat").read_bytes()
    ref = json.loads((tmp_path / "buffer.json").read_text())

    Buf = _messages()
    m = Buf()
    m.ParseFromString(raw)
    assert m.version == ref["version"] == 1 and m.counter == ref["counter"] == 40
    assert m.game_class_name == ref["game_class_name"]
    assert len(m.game_protos) == 40
    for g, r in zip(m.game_protos, ref["games"]):
        assert list(g.actions) == r["actions"]
        np.testing.assert_array_equal(np.array(g.rewards, np.float32), np.array(r["rewards"], np.float32))
        np.testing.assert_array_equal(np.array(g.root_value_targets, np.float32), np.array(r["root_value_targets"], np.float32))
        np.testing.assert_array_equal(np.array(g.maxEntropies, np.float32), np.array(r["max_entropies"], np.float32))
        assert len(g.policy_targets) == len(r["policy_targets"]) == len(r["actions"])
        for pt, rt in zip(g.policy_targets, r["policy_targets"]):
            np.testing.assert_array_equal(np.array(pt.policy, np.float32), np.array(rt, np.float32))
        assert [[int(b) for b in la.legal_action] for la in g.legal_actions] == r["legal_actions"]
        assert len(g.observations) == len(r["observations"]) == len(r["actions"]) + 1
        for o, ro in zip(g.observations, r["observations"]):
            assert o.twoPlayer and o.observationPartSize == 9
            assert _bits(o.observationPartA, 9) == ro["a"] and _bits(o.observationPartB, 9) == ro["b"]
            assert not o.observationPartA.endswith(b"\x00")   # BitSet.toByteArray drops trailing zero bytes
        assert g.tHybrid == r["t_hybrid"] and g.hybrid == bool(r["hybrid"]) and g.count == r["count"]
        assert g.tdSteps == r["td_steps"] and g.trainingEpoch == r["training_epoch"]
        assert g.nextSurpriseCheck == r["next_surprise_check"] and g.surprised == bool(r["surprised"])
        assert g.networkName == r["network_name"] and g.pRandomActionRawCount == r["p_random_action_raw_count"]
        assert np.float32(g.last_value_error) == np.float32(r["last_value_error"])
    # the canonical serialisation of the decoded message is the very byte string the C++ writer produced
    assert m.SerializeToString(deterministic=True) == raw

    z = zipfile.ZipFile(tmp_path / "MuZero-TicTacToe-7-protobuf.zip")
    assert z.namelist() == ["MuZero-TicTacToe-7.dat"]
    assert z.testzip() is None
    assert z.read("MuZero-TicTacToe-7.dat") == raw
