#include "Prismarine/Prismarine.hpp"
#include "Prismarine/Implementations.hpp"   // as the reference: one translation unit of the application includes the bodies
#include "Prismarine/FrameBatch.hpp"
// compile-only: the sharded entry point of the header layer against the C ABI's declarations
void use(psm_dist * d, psm::FrameBatch & b) {
    b.setTile(d, 0, 8, {2, 3, 3, 3, 3, 3, 3, 3});
    b.renderSharded(d, 16, glm::vec3(0, 6, 6), glm::vec3(0, 2, 0), 16, true);
}
int main() { return 0; }
