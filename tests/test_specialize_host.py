"""Run-time specialiser of the fused chain kernel, host side (no GPU): every step code generates source that
compiles for gfx950 with the parity flags (hiprtc cross-compiles without a device), and the generated text is
the straight-line form of the program -- one statement per record, constants read from the argument block."""
import pytest

import kanter_core_amd as kc

ADD, SUB_L, SUB_R, MUL, DIV_L, DIV_R, POW_L, POW_R = range(8)
ADD_INV, SUBL_INV, SUBR_INV, MUL_INV = 10, 11, 12, 13
SAVE_LOAD = 14  # saved = acc; acc = x: a second chain inside the program (csrc/chain_program.h)
SAVED = 254     # operand source index of the saved value (word bits 8-15 = 255)


def word(code, src):
    return code | ((src + 1) << 8)


def test_baseline_program_is_sixteen_straight_line_statements():
    # the 32-node BASELINE graph: 16 records "1 - (acc op B)", op alternating +, *
    words = [word(ADD_INV if i % 2 == 0 else MUL_INV, 1) for i in range(16)]
    src = kc.specialize_compile_check(words, n_in=2, start_src=0, flat=True)
    body = src[src.index('extern "C"'):]
    assert body.count("acc = ") == 17  # start + 16 records
    assert "pp[0].a.c - (acc + in1)" in body and "pp[0].b.c - (acc * in1)" in body and "pp[7].b.c - (acc * in1)" in body
    assert "switch" not in body and "for (" not in body
    assert "__launch_bounds__(256)" in body
    assert "pow_positive" not in src  # only programs with a pow step carry the routine


@pytest.mark.parametrize("flat", [True, False])
def test_every_step_code_compiles(flat):
    plane_codes = [ADD, SUB_L, SUB_R, MUL, DIV_L, DIV_R, POW_L, POW_R, ADD_INV, SUBL_INV, SUBR_INV, MUL_INV]
    const_codes = [ADD, SUB_L, SUB_R, MUL, DIV_L, DIV_R, POW_L, POW_R]
    words = [word(c, i % 4) for i, c in enumerate(plane_codes)] + [word(c, -1) for c in const_codes]
    src = kc.specialize_compile_check(words, n_in=4, start_src=3, flat=flat)
    assert "pow4(acc, in2, pow_tab)" in src and "pow_positive" in src
    assert ("idx / P.row_units" in src) == (not flat)


def test_a_program_that_joins_two_chains():
    """(in0 + in1) * c, then a second chain in2 - in3 inside the same program, then first - second; then the usual invert.
    What config #4's Mix(Add) tree of two unevaluated branches becomes (csrc/runtime.cpp plane_mix / chain_flatten)."""
    words = [word(ADD, 1), word(MUL, -1), word(SAVE_LOAD, 2), word(SUB_L, 3), word(SUBR_INV, SAVED), word(SAVE_LOAD, -1), word(MUL, SAVED)]
    src = kc.specialize_compile_check(words, n_in=4, start_src=0, flat=True)
    body = src[src.index('extern "C"'):]
    assert "acc = (saved0 = acc, in2);" in body and "acc = acc - in3;" in body
    assert "pp[2].a.c - (saved0 - acc)" in body  # saved (first chain) - acc (second chain), then c - that
    assert "acc = (saved0 = acc, f4{ pp[2].b.c" in body and "acc = acc * saved0;" in body


def test_a_joined_chain_with_a_join_of_its_own():
    """in0 op (in1 op (in2 op in3)) with every inner expression a chain of its own: the innermost join happens while two values
    are aside (word bits 16-17 name the saved value)."""
    lvl = lambda w, n: w | (n << 16)  # noqa: E731
    words = [word(ADD, -1),                         # chain 0: in0 + c
             word(SAVE_LOAD, 1), word(MUL, -1),     # saved0 = acc; chain 1: in1 * c
             lvl(word(SAVE_LOAD, 2), 1),            # saved1 = acc; chain 2: in2 ...
             word(SUB_L, 3),                        #   ... - in3
             lvl(word(SUB_R, SAVED), 1),            # chain 1 - chain 2
             word(DIV_R, SAVED)]                    # chain 0 / that
    src = kc.specialize_compile_check(words, n_in=4, start_src=0, flat=True)
    body = src[src.index('extern "C"'):]
    assert "acc = (saved0 = acc, in1);" in body and "acc = (saved1 = acc, in2);" in body
    assert "acc = saved1 - acc;" in body and "acc = saved0 / acc;" in body
    with pytest.raises(kc.TexProError):
        kc.specialize_compile_check([lvl(word(SAVE_LOAD, 1), 3)], n_in=2)  # only three values can be aside
    with pytest.raises(kc.TexProError):
        kc.specialize_compile_check([lvl(word(ADD, 1), 1)], n_in=2)        # a level on a step that names no saved value


def test_constant_start_and_zero_inputs():
    src = kc.specialize_compile_check([word(MUL, -1), word(SUB_R, -1)], n_in=0, start_src=-1)
    assert "P.start_c[b]" in src


def test_invalid_programs_are_refused():
    with pytest.raises(kc.TexProError):
        kc.specialize_compile_check([word(ADD, 2)], n_in=2)      # operand slot out of range
    with pytest.raises(kc.TexProError):
        kc.specialize_compile_check([word(15, 0)], n_in=1)       # unknown code
    with pytest.raises(kc.TexProError):
        kc.specialize_compile_check([word(SAVE_LOAD, SAVED)], n_in=1)  # a second chain cannot start from the saved value
    with pytest.raises(kc.TexProError):
        kc.specialize_compile_check([word(ADD, 0)], n_in=1, start_src=1)


def test_mode_knob_round_trips():
    old = kc.get_specialize()
    try:
        for m in (0, 2, 1):
            kc.set_specialize(m)
            assert kc.get_specialize() == m
        with pytest.raises(kc.TexProError):
            kc.set_specialize(3)
    finally:
        kc.set_specialize(old)
    kc.specialize_wait()
    assert kc.specialize_stats()["compiles_pending"] == 0


# ---- the program inside the integer-ratio up-sampling kernel (upsample_chain.inc embedded as text) ----
@pytest.mark.parametrize("wide", [True, False])
@pytest.mark.parametrize("taps", [1, 3])
def test_upsample_chain_form_compiles(taps, wide):
    # config #2's program: ((A + U) * A) - U with A = input 0 and the resampled operand U = input 1
    words = [word(ADD, 1), word(MUL, 0), word(SUB_L, 1)]
    src = kc.specialize_compile_check_upsample(words, n_in=2, start_src=0, taps=taps, wide=wide)
    body = src[src.index("struct UpProg"):]
    assert "acc = acc + in[1][u];" in body and "acc = acc * in[0][u];" in body and "acc = acc - in[1][u];" in body
    assert "upsample_chain_tile<2, %d, 4, %s, 0x0u>" % (taps, "true" if wide else "false") in body
    assert "kc_upchain_" in body and "switch" not in body


def test_upsample_chain_form_every_code_and_input_count():
    codes = [ADD, SUB_L, SUB_R, MUL, ADD_INV, SUBL_INV, SUBR_INV, MUL_INV]
    for n_in in (1, 2, 3, 4):
        words = [word(c, i % n_in) for i, c in enumerate(codes)] + [word(c, -1) for c in (ADD, SUB_L, SUB_R, MUL)]
        src = kc.specialize_compile_check_upsample(words, n_in=n_in, start_src=n_in - 1)
        assert "f4 acc = in[%d][u];" % (n_in - 1) in src
    src = kc.specialize_compile_check_upsample([word(MUL, 0)], n_in=1, start_src=-1)
    assert "P.start_c[b]" in src


def test_upsample_chain_form_refuses_divide_and_pow():
    for code in (DIV_L, DIV_R, POW_L, POW_R):
        with pytest.raises(kc.TexProError):
            kc.specialize_compile_check_upsample([word(code, 0)], n_in=1)
