// ref_contract.cpp -- the ONE piece of the reference that builds in this image: its data-contract
// header beamformer_coefficient_generator/BeamformerParameters.h (self-contained, no CUDA), included
// from where it lies under the reference root (-I on the command line; nothing is copied).  Prints the
// struct layout and the compile-time constants as JSON: the fixture tests/golden/reference_contract.json
// is this program's output, and tests/test_oracle.py holds the library's and the oracle's defaults to it.
// TEST INFRASTRUCTURE ONLY.  Built into oracle/_ref/ by `make -C oracle ref` when the reference is present.
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "BeamformerParameters.h"

static unsigned bits_of(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, sizeof u);
    return u;
}

int main()
{
    const float sampling_period = SAMPLING_PERIOD;
    const double adc_sample_rate = ADC_SAMPLE_RATE;
    std::printf("{\n");
    std::printf("  \"source\": \"beamformer_coefficient_generator/BeamformerParameters.h\",\n");
    std::printf("  \"sizeof_delay_vals\": %zu,\n", sizeof(struct delay_vals));
    std::printf("  \"offsetof_fDelay_s\": %zu,\n", offsetof(struct delay_vals, fDelay_s));
    std::printf("  \"offsetof_fDelayRate_sps\": %zu,\n", offsetof(struct delay_vals, fDelayRate_sps));
    std::printf("  \"offsetof_fPhase_rad\": %zu,\n", offsetof(struct delay_vals, fPhase_rad));
    std::printf("  \"offsetof_fPhaseRate_radps\": %zu,\n", offsetof(struct delay_vals, fPhaseRate_radps));
    std::printf("  \"COMPLEXITY\": %d,\n", (int)COMPLEXITY);
    std::printf("  \"NR_CHANNELS\": %d,\n", (int)NR_CHANNELS);
    std::printf("  \"NR_POLARIZATIONS\": %d,\n", (int)NR_POLARIZATIONS);
    std::printf("  \"NR_SAMPLES_PER_CHANNEL\": %d,\n", (int)NR_SAMPLES_PER_CHANNEL);
    std::printf("  \"NR_STATIONS\": %d,\n", (int)NR_STATIONS);
    std::printf("  \"NR_BEAMS\": %d,\n", (int)NR_BEAMS);
    std::printf("  \"SAMPLING_PERIOD_f32_bits\": %u,\n", bits_of(sampling_period));
    std::printf("  \"FFT_SIZE\": %d,\n", (int)FFT_SIZE);
    std::printf("  \"ADC_SAMPLE_RATE\": %.1f,\n", adc_sample_rate);
    std::printf("  \"ACCUMULATIONS_BEFORE_NEW_COEFFS\": %d,\n", (int)ACCUMULATIONS_BEFORE_NEW_COEFFS);
    std::printf("  \"NUM_THREADS_PER_BLOCK\": %d,\n", (int)NUM_THREADS_PER_BLOCK);
    std::printf("  \"NUM_ANTBEAMS_PER_BLOCK\": %d,\n", (int)NUM_ANTBEAMS_PER_BLOCK);
    std::printf("  \"INTERNAL_TIME_SAMPLES\": %d\n", (int)INTERNAL_TIME_SAMPLES);
    std::printf("}\n");
    return 0;
}
