"""Host-side model of the two lane geometries of gcr_blend.hip (no GPU): the forward blend's eight 2x4-pixel sub-rows
per wave (lane_geom6), the backward blend's four 4x4-pixel rows (lane_geom), and the map between them that the forward
uses to store its checkpoints where the backward's thread of the same pixel reads them (lane6_to_k7_thread).  The
formulas are the source's, transcribed; what is checked is what the kernels rely on."""


def k7_pixel(tid):
    lane, w = tid & 63, tid >> 6
    row, li = lane >> 4, lane & 15
    bx, by = (w & 1) * 2 + (row & 1), (w >> 1) * 2 + (row >> 1)
    return bx * 4 + (li & 3), by * 4 + (li >> 2)


def k6_pixel(tid):
    lane, w = tid & 63, tid >> 6
    sub, li = lane >> 3, lane & 7
    return (w & 1) * 8 + (sub & 3) * 2 + (li & 1), (w >> 1) * 8 + (sub >> 2) * 4 + (li >> 1)


def lane6_to_k7_thread(tid):
    lane = tid & 63
    sub, li = lane >> 3, lane & 7
    row4 = (sub >> 2) * 2 + ((sub & 3) >> 1)
    li4 = (li >> 1) * 4 + (sub & 1) * 2 + (li & 1)
    return (tid & ~63) + row4 * 16 + li4


def test_each_geometry_owns_every_pixel_of_the_tile_once():
    for f in (k6_pixel, k7_pixel):
        assert sorted(f(t) for t in range(256)) == sorted((x, y) for y in range(16) for x in range(16))


def test_a_wave_is_the_same_quadrant_in_both_geometries():
    for t in range(256):
        (x6, y6), (x7, y7) = k6_pixel(t), k7_pixel(t)
        assert (x6 >> 3, y6 >> 3) == (x7 >> 3, y7 >> 3) == ((t >> 6) & 1, t >> 7)


def test_checkpoint_slot_is_the_backward_thread_of_the_same_pixel():
    seen = set()
    for t in range(256):
        u = lane6_to_k7_thread(t)
        assert k7_pixel(u) == k6_pixel(t)
        seen.add(u)
    assert len(seen) == 256


def test_sub_row_mask_bits():
    """Bit of sub-row s of wave w in gcr_block_mask_2x4 (bit = band * 8 + column of 2 pixels) -- what the list build
    extracts with `mask >> ((w >> 1) * 16 + (w & 1) * 4)` and the bit pattern s < 4 ? s : s + 4."""
    for t in range(256):
        lane, w = t & 63, t >> 6
        sub = lane >> 3
        x, y = k6_pixel(t)
        bit = (y >> 2) * 8 + (x >> 1)
        shift = (w >> 1) * 16 + (w & 1) * 4
        assert bit == shift + (sub if sub < 4 else sub + 4)
        # ... and the 4x4 mask bit the backward uses for the same pixel is the pair-wise OR's target
        assert (y >> 2) * 4 + (x >> 2) == (bit >> 3) * 4 + ((bit & 7) >> 1)
